// solver_subspace.inl -- member functions of Solver<T> (included inside the class body in
// solver.hip): formk (:1681-1908: WN1 kept incrementally, sparse patches, from-scratch Gram fallback,
// host assembly + the two Cholesky factorisations), cmprlb (:1548-1586) and subsm (:2676-2885) --
// W'Z r in closed form or by the cmprlb_wtv pass, the one storing pass of the iteration
// (subsm_update_kernel), the backtracking branch.
  // ==================================================================== formk
  // WN1 from scratch: one masked Gram pass over W (any col; also the fallback when too many
  // variables changed status for the sparse patches)
  int formk_scratch(int col, int head) {
    CHK(commit_pending((const T *)cg, col, head));
    if (wide()) return wide_formk(col, head);
    lbk::launch_formk_gram<T>(q, n, W(), head, col, iwhere);
    const int E = 2 * col * col + col;
    CHK(fetch(E, 0, 0));
    lbh::Mat WN1{snd.data(), 2 * m};
    const int tri = col * (col + 1) / 2;
    for (int i = 0; i < col; ++i)
      for (int j = 0; j <= i; ++j) {
        WN1(i, j) = h_res[i * (i + 1) / 2 + j];                  // Y'ZZ'Y
        WN1(m + i, m + j) = h_res[tri + i * (i + 1) / 2 + j];    // S'AA'S
      }
    for (int i = 0; i < col; ++i)
      for (int j = 0; j < col; ++j) WN1(m + i, j) = h_res[2 * tri + i * col + j];  // L_a + R_z
    return 0;
  }

  // WN1 kept incrementally exactly as the reference does (:1735-1851): shift, new row and
  // column from `nr` (the four sum vectors that rode along in the cmprlb_wtv pass), and
  // patches for the variables that entered / left the free set (sparse signed Gram).
  int formk_incremental(int col, int head, bool updatd, int iupdat, const double *nr, int MCnr,
                        const std::vector<double> *patch = nullptr) {
    const int m2 = 2 * m;
    lbh::Mat WN1{snd.data(), m2};
    const int upcl = updatd ? col - 1 : col;
    const int64_t nchg = nenter_g + (nglob + 1 - ileave_g);
    bool patched = false;
    std::vector<double> P;
    if (patch) {  // (m > 32: the caller's own patch sums, solver_wide.inl)
      if (nchg > 0 && upcl > 0) P = *patch, patched = true;
    } else if (nchg > 0 && upcl > 0 && eager.valid && eager.upcl == upcl && eager.head == head) {
      P = eager.P;  // (came with freev's counts: same kernels, same sums)
      patched = true;
      eager.valid = false;
    } else if (nchg > 0 && upcl > 0) {
      eager.valid = false;
      if (nchg > (int64_t)CHG_CAP) return formk_scratch(col, head);  // whole Gram is cheaper
      // (the list was appended with an atomic counter: put it in ascending order first, so that the
      //  patch sums -- and with them WN1, the subspace step, the whole trajectory -- are
      //  reproducible bit for bit; idx[1] is free here: the walk is over)
      const uint32_t nl = std::min<uint32_t>(chg_local, CHG_CAP);
      const uint32_t *lst = lbk::launch_sort_u32(q, sort_tmp, sort_tmp_bytes, d_chg, idx[1], nl);
      lbk::launch_formk_patch<T>(q, lst, nl, Wc(), head, upcl);
      const int E = 2 * upcl * upcl + upcl;
      CHK(fetch(E, 0, 0));
      P.assign(h_res, h_res + E);
      patched = true;
    }
    if (updatd) {
      if (iupdat > m) {  // shift old part of WN1 (:1736-1744)
        for (int jy = 0; jy < m - 1; ++jy) {
          const int js = m + jy;
          for (int i = 0; i < m - 1 - jy; ++i) {
            WN1(jy + i, jy) = WN1(jy + 1 + i, jy + 1);
            WN1(js + i, js) = WN1(js + 1 + i, js + 1);
          }
          for (int i = 0; i < m - 1; ++i) WN1(m + i, jy) = WN1(m + 1 + i, jy + 1);
        }
      }
      const int nw = col - 1;  // new pair = logical column col-1 (:1746-1793)
      for (int jy = 0; jy < col; ++jy) {
        WN1(nw, jy) = nr[0 * MCnr + jy];          // Y'ZZ'Y row
        WN1(m + nw, m + jy) = nr[1 * MCnr + jy];  // S'AA'S row
        WN1(m + nw, jy) = nr[2 * MCnr + jy];      // L_a row
      }
      for (int i = 0; i < col; ++i) WN1(m + i, nw) = nr[3 * MCnr + i];  // R_z column
    }
    if (patched) {  // :1801-1851 (P = sums over entering rows - sums over leaving rows)
      const int tri = upcl * (upcl + 1) / 2;
      for (int iy = 0; iy < upcl; ++iy)
        for (int jy = 0; jy <= iy; ++jy) {
          WN1(iy, jy) = WN1(iy, jy) + P[iy * (iy + 1) / 2 + jy];
          WN1(m + iy, m + jy) = WN1(m + iy, m + jy) - P[tri + iy * (iy + 1) / 2 + jy];
        }
      for (int is = 0; is < upcl; ++is)
        for (int jy = 0; jy < upcl; ++jy) {
          const double psy = P[2 * tri + is * upcl + jy];
          if (is <= jy)
            WN1(m + is, jy) = WN1(m + is, jy) + psy;
          else
            WN1(m + is, jy) = WN1(m + is, jy) - psy;
        }
    }
    return 0;
  }

  // upper triangle of WN from WN1 and the two Cholesky factorisations (:1856-1906)
  void formk_factor(int col, double theta, int &info) {
    const int m2 = 2 * m;
    lbh::Mat WN{wn.data(), m2}, WN1{snd.data(), m2}, SY{sy.data(), m};
    for (int iy = 0; iy < col; ++iy) {
      const int is = col + iy, is1 = m + iy;
      for (int jy = 0; jy <= iy; ++jy) {
        const int js = col + jy, js1 = m + jy;
        WN(jy, iy) = WN1(iy, jy) / theta;
        WN(js, is) = WN1(is1, js1) * theta;
      }
      for (int jy = 0; jy < iy; ++jy) WN(jy, is) = -WN1(is1, jy);
      for (int jy = iy; jy < col; ++jy) WN(jy, is) = WN1(is1, jy);
      WN(iy, iy) = WN(iy, iy) + SY(iy, iy);
    }
    if (lbh::dpofa(WN, col) != 0) {  // :1880-1884
      info = -1;
      return;
    }
    const int col2 = 2 * col;
    for (int js = col; js < col2; ++js) (void)lbh::dtrsl(WN, col, &WN(0, js), 11);
    for (int is = col; is < col2; ++is)
      for (int js = is; js < col2; ++js)
        WN(is, js) = WN(is, js) + lbh::dot_seq(col, &WN(0, is), &WN(0, js));
    lbh::Mat WN22{&WN(col, col), m2};
    if (lbh::dpofa(WN22, col) != 0) {  // :1902-1906
      info = -2;
      return;
    }
    info = 0;
  }

  int formk(int col, int head, double theta, int &info) {
    CHK(formk_scratch(col, head));
    formk_factor(col, theta, info);
    return 0;
  }

  // ========================================================== cmprlb + subsm
  // coefficients of cmprlb: wa(1:2m) = M c (bmv, :1569) -> a1_j, a2_j = theta * (.) (:1576-1577)
  // (kept in cm_cf / cm_plain: subsm_update_kernel recomputes r from them)
  lbk::Coef cm_cf;
  bool cm_plain = false;
  bool cmprlb_coef(int col, double theta, bool cnstnd, lbk::Coef &cf, bool &plain) {
    std::memset(&cf, 0, sizeof cf);
    plain = !cnstnd && col > 0;
    if (!plain) {
      if (lbh::bmv(m, sy.data(), wt.data(), col, &wa8m[2 * m], &wa8m[0]) != 0) return false;
      for (int j = 0; j < col; ++j) {
        cf.a[j] = wa8m[j];
        cf.a[lbk::MAXM + j] = theta * wa8m[col + j];
      }
    }
    cm_cf = cf, cm_plain = plain;
    return true;
  }

  // W'Z r without a pass over W (cmprlb :1565-1583 folded into subsm :2742-2754).  On the free
  // rows the Cauchy point is x + tsum d with d = -g, so
  //     r = (1 - theta tsum) d + W (M c)   on the free rows Z,   and
  //     W'Z r = (1 - theta tsum) W'Z d + (W'ZZ'W) (M c).
  // W'Z d is the p the walk ends with (it carries W'd over the variables that still move,
  // :1300-1304, :1463-1470); W'ZZ'W is in WN1 and in matupd's matrices:  Y'ZZ'Y = WN1(1:col,1:col),
  // S'ZZ'S = S'S - S'AA'S = Ss - WN1(m+1:,m+1:),  S'ZZ'Y = R_z above the diagonal (WN1), Sy - L_a
  // below it (:1756-1793).  Equal to the sums over the rows up to reassociation -- and to the
  // rounding of z - x, which the row form carries at 1 ulp of x per row: the caller uses this
  // form only when neither the free set nor p is a small remainder of something much larger.
  // ... which is the case while every stored s_i keeps at least 1e-5 of its squared norm on the free
  // rows (variables that sit at a bound do not move: their part of s is zero unless they have
  // just arrived, so a small free SET alone does not make the free PART small)
  // The same kind of difference gives sum_free s_i y_j below the diagonal: Sy(i,j) - L_a(i,j)
  // (total minus active).  Entry by entry that difference may be small against its operands without
  // harm -- what must not drown is its contribution to W'Z r, which is measured against the free
  // norms |Z's_i| |Z'y_j| (Cauchy-Schwarz bounds the exact value by them): the rounding error of the
  // difference, ~eps (|Sy| + |L_a|), has to stay below 1e-5 of that scale.
  bool closed_form_safe(int col) const {
    const double *WN1 = snd.data(), *SS = ss.data(), *SY = sy.data();
    const int m2 = 2 * m;
    const double eps = std::numeric_limits<double>::epsilon();
    for (int i = 0; i < col; ++i) {
      const double tot = SS[(size_t)i + (size_t)i * m];
      const double act = WN1[(size_t)(m + i) + (size_t)(m + i) * m2];
      if (!(tot - act >= 1.0e-5 * tot)) return false;
    }
    for (int i = 1; i < col; ++i) {
      const double ssf = SS[(size_t)i + (size_t)i * m] - WN1[(size_t)(m + i) + (size_t)(m + i) * m2];
      for (int j = 0; j < i; ++j) {
        const double yyf = WN1[(size_t)j + (size_t)j * m2];
        const double tot = SY[(size_t)i + (size_t)j * m], act = WN1[(size_t)(m + i) + (size_t)j * m2];
        const double scale = std::sqrt(std::fabs(ssf) * std::fabs(yyf));
        if (!(eps * (std::fabs(tot) + std::fabs(act)) <= 1.0e-5 * scale)) return false;
      }
    }
    return true;
  }
  void subspace_closed_form(int col, double theta, double *wv) {
    subspace_closed_form(col, theta, cm_cf.a, cm_cf.a + lbk::MAXM, wv);
  }
  // (a1 = (M c)_j, a2 = theta (M c)_{col+j}: cmprlb's coefficients, :1576-1577)
  void subspace_closed_form(int col, double theta, const double *a1, const double *a2, double *wv) {
    const int m2 = 2 * m;
    lbh::Mat WN1{snd.data(), m2}, SY{sy.data(), m}, SS{ss.data(), m};
    const double k1 = 1.0 - theta * gcp.tsum;
    auto YYf = [&](int i, int j) { return i >= j ? WN1(i, j) : WN1(j, i); };
    auto SSf = [&](int i, int j) {
      const double tot = i <= j ? SS(i, j) : SS(j, i);
      const double act = i >= j ? WN1(m + i, m + j) : WN1(m + j, m + i);
      return tot - act;
    };
    auto SYf = [&](int is, int jy) {  // sum_free s_is y_jy
      return is <= jy ? WN1(m + is, jy) : SY(is, jy) - WN1(m + is, jy);
    };
    for (int i = 0; i < col; ++i) {
      double ay = k1 * p_fin[i], as = k1 * (p_fin[col + i] / theta);
      for (int j = 0; j < col; ++j) {
        ay = ay + YYf(i, j) * a1[j] + SYf(j, i) * a2[j];
        as = as + SYf(i, j) * a1[j] + SSf(i, j) * a2[j];
      }
      wv[i] = ay;
      wv[col + i] = theta * as;
    }
  }

  // do_formk: formk is pending for this iteration and col <= 10: its new row/column sums ride
  // along in the cmprlb_wtv pass and the status changes are patched sparsely.
  // closed: no cmprlb pass at all -- new row from the update pass (nrpre, corrected by the
  // walk), W'Z r in closed form.
  int subspace(const T *x, const T *l, const T *u, const int32_t *nbd, const T *g, double theta,
               int col, int head, bool cnstnd, int &iword, int &info, bool do_formk, bool updatd,
               int iupdat, const double *pre, bool closed = false) {
    // cmprlb :1548-1586 (+ W'r of subsm).  `pre` != nullptr: the pass was already launched
    // together with freev's counts (one fetch for both) and its sums are in pre[].
    const int MC = lbk::maxc_for(col);
    const bool newrow = do_formk && updatd;
    const double *res = pre;
    const int ipr = quiet ? -1 : print_level;
    if (closed) {
      lbk::Coef cf;
      bool plain;
      if (!cmprlb_coef(col, theta, cnstnd, cf, plain)) {
        info = -8;
        return 0;
      }
      nclosed++;
    } else if (!pre) {
      lbk::Coef cf;
      bool plain;
      if (!cmprlb_coef(col, theta, cnstnd, cf, plain)) {
        // (the reference would run formk first, :663; either failure refreshes the memory,
        //  after which WN1 is rebuilt from new rows only)
        info = -8;
        return 0;
      }
      CHK(ensure_d(x));
      clk_begin(0);
      lbk::launch_cmprlb_wtv<T>(q, n, x, g, gcp.tsum, iwhere, W(), head, col, theta, cf,
                                newrow ? 1 : 0, r, d, pend);
      clk_end(0);
      CHK(fetch((newrow ? 6 : 2) * MC, 0, 0));
      res = h_res;
    }
    double *wv = &wa8m[0];
    if (!closed) {
      nthreepass++;
      for (int i = 0; i < col; ++i) {
        wv[i] = res[i];
        wv[col + i] = theta * res[MC + i];
      }
    }
    if (do_formk) {
      double nr[4 * lbk::MAXM];
      if (closed) {
        if (newrow) {
          for (int j = 0; j < col; ++j) {
            nr[0 * MC + j] = nrpre.t[0][j] - nrc[0][j];
            nr[1 * MC + j] = nrpre.t[1][j] + nrc[1][j];
            nr[2 * MC + j] = nrpre.t[2][j] + nrc[2][j];
            nr[3 * MC + j] = nrpre.t[3][j] - nrc[3][j];
          }
        }
      } else if (newrow) {
        std::memcpy(nr, res + 2 * MC, sizeof(double) * 4 * MC);
      }
      CHK(formk_incremental(col, head, updatd, iupdat, nr, MC));
      formk_factor(col, theta, info);
      if (info != 0) return 0;
    }
    if (closed && !closed_form_safe(col)) {
      // the free part of some s_i is a tiny remainder of the whole column: S'ZZ'S = S'S - S'AA'S
      // would lose it to cancellation.  W'Z r from a pass over W after all (WN1 is complete: no
      // new-row sums)
      CHK(ensure_d(x));
      clk_begin(0);
      lbk::launch_cmprlb_wtv<T>(q, n, x, g, gcp.tsum, iwhere, W(), head, col, theta, cm_cf, 0, r, d, pend);
      clk_end(0);
      CHK(fetch(2 * MC, 0, 0));
      for (int i = 0; i < col; ++i) {
        wv[i] = h_res[i];
        wv[col + i] = theta * h_res[MC + i];
      }
      closed = false;
      nclosed--, nthreepass++;
    }
    if (closed) subspace_closed_form(col, theta, wv);
    if (ipr >= 99) std::fprintf(rep.out, "\n----------------SUBSM entered-----------------\n\n");  // :2738
    lbh::Mat WN{wn.data(), 2 * m};
    const int col2 = 2 * col;
    info = lbh::dtrsl(WN, col2, wv, 11);
    if (info != 0) return 0;
    for (int i = 0; i < col; ++i) wv[i] = -wv[i];
    info = lbh::dtrsl(WN, col2, wv, 1);
    if (info != 0) return 0;
    lbk::Coef cw;
    std::memset(&cw, 0, sizeof cw);
    for (int j = 0; j < col; ++j) {
      cw.a[j] = wv[j];
      cw.a[lbk::MAXM + j] = wv[col + j];
    }
    // d, t, r get their line-search values in the same pass (see subsm_update_kernel); xp = xcp
    // (:2787) is written out only for state export -- and below if the backtracking branch runs
    if (flags & LBFGSB_F_MIRROR_INDEX) CHK(write_xcp(xp, x, l, u, g));
    // lean: the first trial step is 1 and x = z is stored by the pass, so neither z nor d = x - t
    // is written (5 store streams instead of 7); they stay implicit until ensure_d()
    const bool lean = lean_on && ls_unit_step && (cnstnd || two_pass) && !(flags & LBFGSB_F_MIRROR_INDEX);
    // LBFGSB_F_DEFER_LNSRCH: the pass's four sums (iword, dd_p, dtd, stpmx) are not waited for -- the first
    // trial point x = z is in place, the call returns 'FG_LNSRCH' at once, and the sums come over with
    // the fetch of the NEXT call's first pass (fetch, DEFER_OFF); phase_entry lands them there
    const bool defer = defer_on && ls_unit_step && !(flags & (LBFGSB_F_MIRROR_INDEX | LBFGSB_F_PARALLEL_GCP)) &&
                       ipr < 99;
    clk_begin(2);
    // (ping-pong buffers: no t = x, r = g copies -- the roles change below -- and the trial point
    //  goes to the other x buffer, which is where the pending pair's t is read from, row by row)
    q.res_off = defer ? DEFER_OFF : 0;
    // (deferred: nobody waits for these sums in this call -- they go to a partial-sum matrix of their own and
    //  their finalize rides with the next launch's, usually the caller's objective or the update pass)
    if (defer && fold_fin) q.part_sel = 2, q.hold_fin = true;
    seg(4);
    if (t_mid0 > 0.0) t_mid += now_s() - t_mid0, n_mid++, t_mid0 = 0.0;
    lbk::launch_subsm_update<T>(q, n, gcp.tsum, lean ? (T *)nullptr : z, r, pp ? (T *)nullptr : r, lk(l), uk(u),
                                nbk(), iwhere, x, g, Wc(), head, col, theta, cm_cf, cw, lean ? (T *)nullptr : d,
                                pp ? (T *)nullptr : t, ls_unit_step ? xmut : nullptr, ls_do_stpmx ? 1 : 0,
                                pend, d_src(), ub_mask);
    q.res_off = 0, q.part_sel = 0;
    clk_end(2);
    pend.on = 0, pend.impl = 0;  // the pass stored the pair into its W slot
    d_impl = z_in_x = lean;
    z_valid = !lean;
    if (lean) x_lean = xmut;
    if (pp) t = const_cast<T *>(x), r = const_cast<T *>(g);  // t = x, r = g (:2235-2236) as a change of roles
    sub_cw = cw, sub_theta = theta, sub_col = col, sub_head = head;
    if (defer) {
      defer_live = true, ls.deferred = true;
      ndeferred++;
      return 0;
    }
    CHK(fetch(3, 1, 0));
    return subspace_land(x, l, u, nbd, g, h_res, iword, info);
  }

  // what follows the storing pass's sums (in the same call, or -- deferred -- in the next one): the
  // line-search set-up values, and the backtracking branch (:2830-2879) when the projected step points uphill
  lbk::Coef sub_cw;
  double sub_theta = 1.0;
  int sub_col = 0, sub_head = 1;
  int subspace_land(const T *x, const T *l, const T *u, const int32_t *nbd, const T *g, const double *R,
                    int &iword, int &info) {
    const int ipr = quiet ? -1 : print_level;
    const double theta = sub_theta;
    const int col = sub_col, head = sub_head;
    const lbk::Coef &cw = sub_cw;
    if (R[0] >= 1.0e29)  // (k_subsm.hip: a free row outside its tile's front run -- the layout was not re-sorted)
      return fail(LBFGSB_E_STATE, "storing pass: the tile-local layout of W is older than the free set");
    iword = R[0] > 0.0 ? 1 : 0;
    const double dd_p = R[1];
    ls.ready = true;
    ls.x_is_z = ls_unit_step;
    ls.gd = dd_p;
    ls.dtd = R[2];
    ls.stpmx = R[3];
    if (iword == 0 || dd_p <= 0.0) {  // :2820, :2828
      if (ipr >= 99) std::fprintf(rep.out, "\n----------------exit SUBSM --------------------\n\n");  // :2883
      return 0;
    }
    ls.ready = false;  // z changes below: lnsrlb_begin redoes d, t, r
    d_impl = z_in_x = false;  // (and the backtracking kernel writes all of z)
    if (ls.x_is_z) {   // ... from the iterate itself, which the pass above saved in t
      // (ping-pong buffers: the trial point went to the other buffer, x still is the iterate)
      if (!pp) HIPCHK(hipMemcpyAsync(xmut, t, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, stream));
      ls.x_is_z = false;
    }
    if (rep.out && !quiet && print_level >= 0) {
      std::fprintf(rep.out, " Positive dir derivative in projection \n");
      std::fprintf(rep.out, " Using the backtracking step \n");
    }
    // xp = xcp and the Newton direction as vectors (the direction goes to tbrk, which the
    // cursor-based cauchy_finish_kernel has read by then)
    if (!(flags & LBFGSB_F_MIRROR_INDEX)) CHK(write_xcp(xp, x, l, u, g));
    lbk::launch_subsm_dir<T>(q, n, xp, iwhere, x, g, W(), head, col, theta, cm_cf, cw, tbrk);
    tbrk_valid = false;
    lbk::launch_subsm_alpha<T>(q, n, xp, tbrk, l, u, nbd, iwhere);
    CHK(fetch(0, 1, 0));
    const double alpha = std::min(1.0, h_res[0]);
    int64_t ibd = -1;
    if (alpha < 1.0) {
      lbk::launch_subsm_argalpha<T>(q, n, row0, xp, tbrk, l, u, nbd, iwhere, alpha);
      CHK(fetch(0, 1, 0));
      ibd = (int64_t)h_res[0];
    }
    lbk::launch_subsm_backtrack<T>(q, n, row0, z, xp, tbrk, l, u, iwhere, alpha, ibd);
    if (ipr >= 99) std::fprintf(rep.out, "\n----------------exit SUBSM --------------------\n\n");
    (void)info;
    return 0;
  }

  // line-search set-up values when they were produced by the subsm pass
  struct LsOut {
    bool ready = false;
    bool deferred = false;  // the storing pass ran, its sums are still on the device (defer_live)
    bool x_is_z = false;  // the pass already stored the first trial point x = z
    double gd = 0, dtd = 0, stpmx = 0;
  } ls;
  bool ls_do_stpmx = false;
  bool ls_unit_step = false;  // the first trial step of this iteration's line search is 1
  T *xmut = nullptr;          // the caller's x of this call
  // sums of a cmprlb_wtv pass that was launched together with freev's counts
  double pre_res[6 * lbk::MAXM];
  bool pre_valid = false;
  // ---- two-pass iteration (col <= 10): formk's new row rides in the update pass with the
  //      pre-walk free set, the walk corrects it for the rows it fixes, and W'Z r follows in
  //      closed form from the walk's p and WN1 (subspace_closed_form) -- no cmprlb pass ----
  bool two_pass = true;  // (option "two_pass")
  bool wide_incr_on = true;  // (option "wide_incr": m > 32 keeps WN1 incrementally)
  // (every col the fused kernels take; beyond 21 stored pairs the update pass has no registers for the 4 col + 4
  //  extra sums and runs as two launches over half of the columns each, k_update.hip "the split pass";
  //  option "two_pass_maxcol" lowers the limit, for measurements: 20 = round 3's three passes at col > 20)
  int two_pass_maxcol = lbk::MAXM;
  bool exact_always = false;  // (option "exact_always": every walk in the reference's heap order)
  bool defer_on = false;      // (LBFGSB_F_DEFER_LNSRCH / option "defer_lnsrch")
  bool fold_fin = true;       // (option "fold_finalize": reductions nobody waits for yet park their finalize)
  double defer_f0 = 0.0;      // f at the iterate of a deferred line-search set-up
  int64_t ndeferred = 0, nredo = 0;  // set-ups whose sums were deferred / of those, requests that had to be re-issued

  // lbfgsb_hip_set_option: measurement / test switches of THIS context (include/lbfgsb_hip.h)
  int set_option(const char *name, double v) override {
    const std::string k = name ? name : "";
    const auto flag = [&](bool &dst) -> int {
      if (v != 0.0 && v != 1.0) return fail(LBFGSB_E_ARG, "set_option: " + k + " takes 0 or 1");
      dst = v != 0.0;
      return 0;
    };
    const auto in_range = [&](int lo, int hi, int &dst) -> int {
      if (!(v >= lo && v <= hi) || v != std::floor(v))
        return fail(LBFGSB_E_ARG, "set_option: " + k + " out of range");
      dst = (int)v;
      return 0;
    };
    if (k == "compact_w") {  // the two passes over W on the tile-local free-row layout (fp64, m <= 10)
      const int rc = in_range(0, 2, cw_mode);
      if (rc) return rc;
      cw_on = cw_mode != 0;
      if (!cw_on) (void)W();  // (back to natural order for the kernels that will run now)
      return 0;
    }
    if (k == "compact_min_rows") {
      if (!(v >= -1.0 && v <= 9.0e15) || v != std::floor(v)) return fail(LBFGSB_E_ARG, "set_option: compact_min_rows out of range");
      cw_min_rows = (int64_t)v;
      return 0;
    }
    if (k == "compact_policy") return in_range(0, 2, cw_policy);
    if (k == "two_pass") return flag(two_pass);
    if (k == "two_pass_maxcol") return in_range(0, lbk::MAXM, two_pass_maxcol);
    if (k == "lean") return flag(lean_on);
    if (k == "spec_capture") return flag(spec_on);
    if (k == "exact_always") return flag(exact_always);
    if (k == "defer_lnsrch") return flag(defer_on);
    if (k == "spin") {
      const int rc = flag(spin_on);
      q.fin_publish = spin_on && !comm;
      return rc;
    }
    if (k == "fold_finalize") return flag(fold_fin);
    if (k == "eager_patch") return flag(eager_on);
    if (k == "spec_freev") return flag(spec_freev_on);
    if (k == "spec_trial2") return flag(spec_trial2_on);
    if (k == "skip_reuse") return flag(skip_reuse_on);
    if (k == "wide_incr") return flag(wide_incr_on);
    if (k == "wide_fused") return flag(wide_fused_on);
    if (k == "wide_closed") return flag(wide_closed_on);
    if (k == "wide_tail") return flag(wide_tail_on);
    if (k == "wide_one") return flag(wide_one_on);
    if (k == "nt") return flag(q.nt);
    if (k == "win_slack") {  // the walk's first window asks (1 + win_slack) x as far ahead as it needs
      if (!(v >= 0.0 && v <= 8.0)) return fail(LBFGSB_E_ARG, "set_option: win_slack must be in [0, 8]");
      win_slack = v;
      return 0;
    }
    if (k == "pg_min") {
      if (!(v >= 0.0)) return fail(LBFGSB_E_ARG, "set_option: pg_min must be >= 0");
      PG_MIN = v;
      return 0;
    }
    if (k == "uniform_bounds") {
      const int rc = flag(ub_on);
      if (!ub_on) {
        if (ub_mask & lbk::UB_DICT) nbd8_src = nullptr;  // (the plain nbd bytes again at the next call)
        ub_mask = 0;
      }
      return rc;
    }
    if (k == "dict_bounds") return flag(dict_on);            // (before START)
    if (k == "bounds_check") return in_range(0, 1 << 20, bcheck_every);
    if (k == "wgrid") return in_range(0, lbk::MAX_BLOCKS - 1, q.tune.wgrid);
    if (k == "pipe") return in_range(-1, 1, q.tune.pipe);
    if (k == "pair") return in_range(0, 2, q.tune.pair);
    if (k == "gram_rows") return in_range(0, 1, q.tune.gram_rows);
    if (k == "pair_cw") return in_range(0, 1, q.tune.pair_cw);
    if (k == "split") {  // 20 (default: parts of <= 16 columns beyond 20 old pairs) or 10 (parts of <= 10 beyond 10)
      if (v != 10.0 && v != 20.0) return fail(LBFGSB_E_ARG, "set_option: split takes 10 or 20");
      if (v == 10.0 && lbk::split_parts(m - 1, 10) > lbk::SPLIT_MAXPARTS)
        return fail(LBFGSB_E_ARG, "set_option: split = 10 needs more parts than the split pass has (m too large)");
      q.tune.split_from = (int)v, q.tune.split_cols = v == 10.0 ? 10 : 16;
      return 0;
    }
    return fail(LBFGSB_E_ARG, "set_option: unknown option '" + k + "'");
  }
  // update_scan_kernel's NEWROW flag for the pass that forms pair number `colnew`
  // (m > 32: every col while the update pass stays fused in front of the unfused subspace steps)
  int nr_flag(int colnew) const { return two_pass && (colnew <= two_pass_maxcol || wide_fused()) ? 1 : 0; }
  // m > 32: matupd's, cauchy's and formk's sums from the (split) update pass -- one pass over W instead of
  // five; the subspace steps stay the unfused ones (solver_wide.inl).  Option "wide_fused" = 0: all unfused
  bool wide_fused_on = true, wide_closed_on = true;  // ("wide_closed": W'Z r in closed form, one axpy pass)
  bool wide_tail_on = true;  // ("wide_tail": cmprlb's start and subsm's tail folded into that pass's first / last tile)
  bool spec_trial2_on = true;  // ("spec_trial2": the second trial of a line search is evaluated by the update pass too)
  bool wide_one_on = true;   // ("wide_one": that pass as ONE launch for col <= 96, the pending pair committed by it)
  bool wide_fused() const { return wide() && wide_fused_on && two_pass; }
  struct NewRow {
    bool valid = false;
    int col = 0;
    // logical columns 0..col-1: Y'ZZ'Y row, S'AA'S row, L_a row, R_z column
    std::vector<double> t[4] = {std::vector<double>(LBFGSB_MAX_M, 0.0), std::vector<double>(LBFGSB_MAX_M, 0.0),
                                std::vector<double>(LBFGSB_MAX_M, 0.0), std::vector<double>(LBFGSB_MAX_M, 0.0)};
  } nrpre;
  // what the walk's fixed rows take from / add to them
  std::vector<double> nrc[4] = {std::vector<double>(LBFGSB_MAX_M, 0.0), std::vector<double>(LBFGSB_MAX_M, 0.0),
                                std::vector<double>(LBFGSB_MAX_M, 0.0), std::vector<double>(LBFGSB_MAX_M, 0.0)};
  void nrc_clear() {
    for (auto &v : nrc) std::fill(v.begin(), v.end(), 0.0);
  }
  std::vector<double> p_fin = std::vector<double>(2 * (size_t)LBFGSB_MAX_M, 0.0);
  double p_ini_max = 0.0;
  bool closed_ok = false;    // this call's cauchy left everything the closed form needs
  int64_t nclosed = 0, nthreepass = 0;

