#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS use of the gfx950 code objects inside the built library.

    python profiles/scripts/kernel_resources.py [lib-or-object ...] [--csv OUT] [--isa-dir DIR]

Reads the `.hip_fatbin` section of each file (default: lbfgsb_amd/liblbfgsb_hip.so), unbundles the
clang offload bundle by hand (no perl roc-obj tools needed), and prints for every kernel the fields of
its AMDHSA metadata note: vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, private_segment_fixed_size
(scratch bytes per lane), group_segment_fixed_size (LDS bytes), max_flat_workgroup_size -- plus the waves
per SIMD the unified 512-entry register file of CDNA4 allows.  tests/test_code_objects_cpu.py asserts
scratch == 0 for every kernel a default context can launch.
"""
import argparse
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def fatbin_sections(path):
    """the raw bytes of .hip_fatbin (a concatenation of offload bundles)"""
    with tempfile.NamedTemporaryFile(suffix=".fatbin") as tf:
        subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, tf.name])
        return open(tf.name, "rb").read()


def unbundle(blob):
    """yield (triple, code_object_bytes) for every device entry of every bundle in blob"""
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        base = pos
        (nent,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(nent):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            p += 24
            triple = blob[p:p + tl].decode()
            p += tl
            if size and "amdgcn" in triple:
                yield triple, blob[base + off: base + off + size]
        pos = base + len(MAGIC)


def kernels_of(code_object):
    """parse `llvm-readelf --notes` (AMDHSA metadata YAML) into a list of dicts"""
    with tempfile.NamedTemporaryFile(suffix=".co") as tf:
        tf.write(code_object)
        tf.flush()
        txt = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", tf.name], text=True)
    out, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count":          # first key of a kernel entry in the note (alphabetical)
            cur = {}
            out.append(cur)
        if cur is not None and k in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                     "private_segment_fixed_size", "group_segment_fixed_size",
                                     "max_flat_workgroup_size", "name", "symbol", "uses_dynamic_stack"):
            cur[k] = v
    return [k for k in out if "name" in k]


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), text=True, capture_output=True)
    return p.stdout.splitlines()


def waves_per_simd(vgpr, agpr):
    # gfx950: 512 unified VGPR/AGPR entries per lane and SIMD, allocation granule 8, at most 8 waves;
    # the note's vgpr_count is the unified total (architectural VGPRs + AGPRs)
    tot = ((int(vgpr) + 7) // 8) * 8
    return max(1, min(8, 512 // max(tot, 1)))


def collect(paths):
    rows = []
    for path in paths:
        for triple, co in unbundle(fatbin_sections(path)):
            if "gfx950" not in triple:
                continue
            ks = kernels_of(co)
            for k, dn in zip(ks, demangle([k["name"] for k in ks])):
                rows.append(dict(file=os.path.basename(path), kernel=dn, vgpr=int(k.get("vgpr_count", 0)),
                                 agpr=int(k.get("agpr_count", 0)), sgpr=int(k.get("sgpr_count", 0)),
                                 vgpr_spill=int(k.get("vgpr_spill_count", 0)),
                                 sgpr_spill=int(k.get("sgpr_spill_count", 0)),
                                 scratch=int(k.get("private_segment_fixed_size", 0)),
                                 dyn_stack=k.get("uses_dynamic_stack", "false"),
                                 lds=int(k.get("group_segment_fixed_size", 0)),
                                 wg=int(k.get("max_flat_workgroup_size", 0)), code_object=co))
    for r in rows:
        r["waves_per_simd"] = waves_per_simd(r["vgpr"], r["agpr"])
    return rows


def short(name):
    """'void lbk::foo<double, 10, true>(long, ...)' -> 'foo<double, 10, true>'"""
    s = re.sub(r"^void\s+", "", name)
    depth, cut = 0, len(s)
    for i, c in enumerate(s):
        if c == "<":
            depth += 1
        elif c == ">":
            depth -= 1
        elif c == "(" and depth == 0:
            cut = i
            break
    s = s[:cut].replace("lbk::", "")
    if len(s) > 160:        # (rocPRIM's sort kernels: kilobytes of template arguments)
        s = s[:150] + "...>"
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--csv")
    ap.add_argument("--isa-dir", help="also write the disassembly of every code object here")
    ap.add_argument("--only-scratch", action="store_true")
    a = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    files = a.files or [os.path.join(root, "lbfgsb_amd", "liblbfgsb_hip.so")]
    rows = collect(files)
    rows.sort(key=lambda r: (short(r["kernel"])))
    hdr = "kernel,vgpr,agpr,sgpr,vgpr_spill,scratch_bytes,lds_bytes,waves_per_simd"
    lines = [hdr]
    for r in rows:
        if a.only_scratch and not (r["scratch"] or r["vgpr_spill"]):
            continue
        lines.append('"%s",%d,%d,%d,%d,%d,%d,%d' % (short(r["kernel"]), r["vgpr"], r["agpr"], r["sgpr"],
                                                   r["vgpr_spill"], r["scratch"], r["lds"], r["waves_per_simd"]))
    text = "\n".join(lines) + "\n"
    if a.csv:
        open(a.csv, "w").write(text)
    else:
        sys.stdout.write(text)
    nscr = sum(1 for r in rows if r["scratch"])
    sys.stderr.write("%d kernels, %d with scratch, fatbin %.1f MB\n" %
                     (len(rows), nscr, sum(len(c) for c in {id(r["code_object"]): r["code_object"] for r in rows}.values()) / 1e6))
    if a.isa_dir:
        os.makedirs(a.isa_dir, exist_ok=True)
        seen = {}
        for r in rows:
            seen.setdefault(id(r["code_object"]), (r["file"], r["code_object"]))
        for k, (fn, co) in enumerate(seen.values()):
            p = os.path.join(a.isa_dir, "%s_%d.co" % (fn, k))
            open(p, "wb").write(co)
            with open(p[:-3] + ".s", "w") as f:
                subprocess.check_call([LLVM + "/llvm-objdump", "-d", "--mcpu=gfx950", p], stdout=f)


if __name__ == "__main__":
    main()
