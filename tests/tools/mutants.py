"""Mutation check of the GPU test suite: deliberately broken builds of the library must be caught.

    python tests/tools/mutants.py build          (here, CPU: one full build per mutant under build/)
    bash   tests/tools/mutants_run.sh            (GPU box: the -m gpu suite with -x against every mutant)

Each mutant is ONE small textual change of csrc/ebm_kernels.hip (or csrc/ebm_runtime.hip) — a sign, a dropped select, a broken halo — of the
kind a transcription error would be.  The runner records the first test that fails for each; a mutant that the whole
suite lets pass is a hole in the suite.  Nothing here is product code: the mutants live under build/ only."""
import os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MUTANTS = [
    ("flux_metric_sign", "return ieee_div((1.0 - xx * xx) * dT, hi_x - lo_x);", "return ieee_div((1.0 + xx * xx) * dT, hi_x - lo_x);"),
    ("uniform_stencil_skips_equatorward_neighbour", "y = (k > 0) ? y + g0 * tbm : y;", "y = (k > 1) ? y + g0 * tbm : y;"),
    ("water_temp_keeps_nan", "return __builtin_isnan(tw) ? 0.0 : tw;", "return tw;"),
    ("t0_rhs_forcing_sign", "return -((p.ai * S - p.A) + dif + f);", "return -((p.ai * S - p.A) + dif - f);"),
    ("lead_ring_radius_sign", "const double Dr = Dk + p.two_rl;", "const double Dr = Dk - p.two_rl;"),
    ("ice_enthalpy_not_clamped", "const double cEi = jl_clamp(rEi, -INFINITY, 0.0);", "const double cEi = rEi;"),
    ("lateral_growth_guard_inverted", "if (hk == 0.0) lat_grow = 0.0;", "if (hk != 0.0) lat_grow = 0.0;"),
    ("welding_quadratic", "const double weld = p.c_weld * ph * (Dk * Dk * Dk);", "const double weld = p.c_weld * ph * (Dk * Dk);"),
    ("concentration_not_capped", "if (phi_n > 1.0) phi_n = 1.0;", ""),
    ("halo_left_is_own_value", "left = t > 0 ? l : 0.0;", "left = t > 0 ? first : 0.0;"),
    ("back_substitution_drops_left_interface", "x[i] = __builtin_fma(-cp[i], x[i + 1], __builtin_fma(lp[i], L, dp[i]));",
     "x[i] = __builtin_fma(-cp[i], x[i + 1], dp[i]);"),
    # (an EQUIVALENT mutant, kept as a control: with E == +-0 both forms give +-0 — the suite cannot and need not see it)
    ("classic_zero_enthalpy_is_ice", "const double Tk = bool_mul(ieee_div(Ek, p.cw), Ek >= 0.0)", "const double Tk = bool_mul(ieee_div(Ek, p.cw), Ek > 0.0)"),
    # second batch
    ("schedule_ramp_ignores_its_start", "else if (tyear < d2) v = base + up * (tyear - d1);", "else if (tyear < d2) v = base + up * tyear;"),
    ("floe_number_linear_in_D", "double n = ieee_div(ph, alpha * (Dk * Dk));", "double n = ieee_div(ph, alpha * Dk);"),
    ("olr_slope_sign", "const double L = p.A + p.B * (tb - Tm);", "const double L = p.A - p.B * (tb - Tm);"),
    ("water_albedo_sign", "const double sol_w = 0.0 + (p.a0 - p.a2 * (xk * xk)) * S;", "const double sol_w = 0.0 + (p.a0 + p.a2 * (xk * xk)) * S;"),
    ("lateral_flux_without_pi", "double Flat = ieee_div(ph * hk * Lf * wl * M_PI, alpha * Dk);", "double Flat = ieee_div(ph * hk * Lf * wl, alpha * Dk);"),
    ("surplus_of_ice_not_given_to_water", "const double Ew_n = cEw + psiEidt;", "const double Ew_n = cEw;"),
    ("lead_share_added_not_subtracted", "const double Qp = psi - Ql;", "const double Qp = psi + Ql;"),
    ("lateral_melt_sign", "const double lat_melt = p.c_latmelt * wl;", "const double lat_melt = -(p.c_latmelt * wl);"),
    ("thickness_tendency_sign", "double rh = hk + (p.c_ht * Fvi) * dt;", "double rh = hk - (p.c_ht * Fvi) * dt;"),
    ("new_ice_thickness_uses_Dmin", "double h_n = div_with_rcp(n * rh + dn * p.hmin, total, rtotal);", "double h_n = div_with_rcp(n * rh + dn * p.Dmin, total, rtotal);"),
    ("mean_temperature_uses_old_concentration", "o.q[Q_T] = Ti * phi_n + (1.0 - phi_n) * Tw;", "o.q[Q_T] = Ti * ph + (1.0 - ph) * Tw;"),
    ("insolation_seasonal_sign", "return p.S0 - p.S1 * xk * ct - p.S2 * (xk * xk);", "return p.S0 + p.S1 * xk * ct - p.S2 * (xk * xk);"),
    ("t0_conduction_ignores_hmin", "return __builtin_fma(p.k, fast_rcp((hk == 0.0) ? p.hmin : hk), p.B);", "return __builtin_fma(p.k, fast_rcp(hk), p.B);"),
    # (the second EQUIVALENT control: at T0 == Tm exactly both branches of the piecewise-linear system agree)
    ("active_set_includes_melting_point", "snew |= (xs[i] < 0.0) ? (1u << i) : 0u;", "snew |= (xs[i] <= 0.0) ? (1u << i) : 0u;"),
    ("active_rows_ignore_concentration", "g[i] = ((smask >> i) & 1u) ? ph[i] : 0.0;", "g[i] = ((smask >> i) & 1u) ? 1.0 : 0.0;"),
    ("classic_ocean_heat_flux_sign", "Ek = Ek + p.dt * (Cc - p.M * Tk + p.Fb);", "Ek = Ek + p.dt * (Cc - p.M * Tk - p.Fb);"),
    ("extension_matrix_diagonal_sign", "rb[i] = 1.0 + p.theta_imex * (tlo[i] + tup[i]);", "rb[i] = 1.0 - p.theta_imex * (tlo[i] + tup[i]);"),
    # third batch: the host runtime (tables, time bookkeeping, savesol!) and the small kernels; a 4-tuple names the file
    ("uniform_table_metric_linear", "lam[i - 1] = (1.0 - xb * xb) / (dx * dx);", "lam[i - 1] = (1.0 - xb) / (dx * dx);", "ebm_runtime.hip"),
    ("polar_ghost_cell_misplaced", "double xp = k < nx - 1 ? x[k + 1] : 2.0 - x[nx - 1];", "double xp = k < nx - 1 ? x[k + 1] : 1.0 - x[nx - 1];", "ebm_runtime.hip"),
    ("solver_table_uses_wrong_spacing", "double l = p.D * g1[k] / (g3[k] * g4[k]);", "double l = p.D * g1[k] / (g2[k] * g4[k]);", "ebm_runtime.hip"),
    ("classic_ghost_diagonal_sign", "kdiag[k] = one - (dtD * g1[k]) / p.cg;", "kdiag[k] = one + (dtD * g1[k]) / p.cg;", "ebm_runtime.hip"),
    ("model_time_at_step_start", "return nt > 0.0 ? (double)(2 * step + 1) / (2.0 * nt) : 0.0;", "return nt > 0.0 ? (double)(2 * step) / (2.0 * nt) : 0.0;", "ebm_runtime.hip"),
    ("lastonly_keeps_one_step_too_many", "const bool want_raw = stage && (!lastonly || tinx > total - nt);", "const bool want_raw = stage && (!lastonly || tinx >= total - nt);", "ebm_runtime.hip"),
    ("winter_snapshot_one_step_late", "if (ti == winter_inx) {", "if (ti == winter_inx + 1) {", "ebm_runtime.hip"),
    ("annual_mean_divides_by_nt_minus_1", "m.x = s.x / nt;", "m.x = s.x / (nt - 1.0);"),
    ("annual_sum_not_restarted", "z.x = 0.0;", "z.x = s.x;"),
    ("hemispheric_mean_without_the_half", "terms[i] = ieee_div((v[i] + v[i + 1]) * (x[i + 1] - x[i]), 2.0);", "terms[i] = (v[i] + v[i + 1]) * (x[i + 1] - x[i]);"),
    # fourth batch: code that only the fused-K kernels, savesol!-in-the-step, the extension and the classic solve run
    # (a seventh mutant of this batch removed the barrier that closed the classic kernel's K-step loop and SURVIVED: the
    # barrier was redundant — see the comment there — and is gone)
    ("fused_steps_all_use_the_first_scalars", "const StepSched sc = a.sched[a.slot + step];", "const StepSched sc = a.sched[a.slot];"),
    ("fused_diagnostics_of_the_first_step", "const bool diag = a.write_diag && step == nloop - 1;", "const bool diag = a.write_diag && step == 0;"),
    ("running_sum_drops_every_second_cell", "s.y = s.y + x1;", "s.y = s.y + x0;"),
    ("snapshot_ring_ignores_its_offset", "EBM_STORE2(a.stage + (size_t)v * a.stage_var_stride + a.stage_offset + col_off + kp, d);",
     "EBM_STORE2(a.stage + (size_t)v * a.stage_var_stride + col_off + kp, d);"),
    ("extension_lower_diagonal_sign", "ra[i] = -(p.theta_imex * tlo[i]);", "ra[i] = (p.theta_imex * tlo[i]);"),
    ("classic_ghost_layer_uses_this_steps_sun", "const double S_ip1 = Sb[i] - (p.S1 * ct_next) * xk[i];", "const double S_ip1 = Sb[i] - (p.S1 * ct) * xk[i];"),
    # fifth batch: the solver's own algebra, the store paths of the two launch geometries, the iteration's control flow
    ("column_offsets_all_take_the_first", "double f = a.fcol ? ft + a.fcol[col] : ft;", "double f = a.fcol ? ft + a.fcol[0] : ft;"),
    ("partition_interface_row_sign", "const double RB = __builtin_fma(ce, vn, __builtin_fma(-ae, cp[C - 2], be));", "const double RB = __builtin_fma(ce, vn, __builtin_fma(ae, cp[C - 2], be));"),
    ("second_level_left_coupling_sign", "lq[i] = -(a2[i] * lq[i - 1]) * w;", "lq[i] = (a2[i] * lq[i - 1]) * w;"),
    ("cyclic_reduction_lower_sign", "const double nqa = -(qa * am) * r;", "const double nqa = (qa * am) * r;"),
    ("t0_diagonal_without_olr", "return __builtin_fma(p.k, fast_rcp((hk == 0.0) ? p.hmin : hk), p.B);", "return __builtin_fma(p.k, fast_rcp((hk == 0.0) ? p.hmin : hk), 0.0);"),
    ("newton_stops_after_one_iteration", "} while (again && it < kMaxNewton);", "} while (false);"),
    ("warm_start_forgotten", "cmask[tl] = (unsigned short)smask;                // new warm start", "cmask[tl] = (unsigned short)0;                    // new warm start"),
    ("two_cell_geometry_swaps_h_and_D", "EBM_PUT(S_Ei, Q_Ei) EBM_PUT(S_Ew, Q_Ew) EBM_PUT(S_h, Q_h) EBM_PUT(S_D, Q_D) EBM_PUT(S_phi, Q_phi)",
     "EBM_PUT(S_Ei, Q_Ei) EBM_PUT(S_Ew, Q_Ew) EBM_PUT(S_h, Q_D) EBM_PUT(S_D, Q_h) EBM_PUT(S_phi, Q_phi)"),
    ("parked_first_pair_takes_D_for_h", "sEw[0] = v0 ? o[0].q[Q_h] : 0.0;  sEw[T] = v1 ? o[1].q[Q_h] : 0.0;", "sEw[0] = v0 ? o[0].q[Q_D] : 0.0;  sEw[T] = v1 ? o[1].q[Q_h] : 0.0;"),
    ("classic_ghost_diagonal_ignores_the_melting_mask", "const double q = bool_mul(bool_mul(ieee_div(p.dc, den), T0 < 0.0), Ek < 0.0);", "const double q = bool_mul(ieee_div(p.dc, den), Ek < 0.0);"),
    ("classic_surface_temperature_sign", "const double T0 = ieee_div(Cc, p.M - ieee_div(p.kLf, Ek));", "const double T0 = ieee_div(Cc, p.M + ieee_div(p.kLf, Ek));"),
    # sixth batch (round 3): the private store layout of the diagnostic fields, validity tracking, launch chains, the pinned
    # ring, model time across integrate calls, the zonal sweep and its tables
    ("unsplit_writes_the_first_pair_twice", "*reinterpret_cast<double2 *>(f + 4 * t + 2) = p1;", "*reinterpret_cast<double2 *>(f + 4 * t + 2) = p0;"),
    ("diagnostic_pairs_stored_on_top_of_each_other", "const unsigned ks = (unsigned)(j * 2 * T + 2 * t);", "const unsigned ks = (unsigned)(2 * t);"),
    ("diagnostic_layout_flag_never_set", "if (write_diag && h->model == EBM_MODEL_MIZ) h->diag_split = h->cfg.cells == 4;",
     "if (write_diag && h->model == EBM_MODEL_MIZ) h->diag_split = false;", "ebm_runtime.hip"),
    ("fields_never_go_stale", "    h->epoch += nsteps;", "    h->epoch += 0;", "ebm_runtime.hip"),
    ("classic_kernel_ignores_its_column_offset", "const int T = blockDim.x, t = threadIdx.x, col = a.col0 + (int)blockIdx.x;",
     "const int T = blockDim.x, t = threadIdx.x, col = (int)blockIdx.x;"),
    ("launch_chains_never_joined", "        (void)hipStreamWaitEvent(h->stream, h->ev_join, 0);", "", "ebm_runtime.hip"),
    ("second_chain_does_not_wait_for_earlier_work", "if (e == hipSuccess) e = hipStreamWaitEvent(h->stream2, h->ev_fork, 0);", "", "ebm_runtime.hip"),
    ("zonal_last_row_coefficient_sign", "            f = -a * ee[i];", "            f = a * ee[i];"),
    ("zonal_back_substitution_drops_the_wrap_term", "const double U = __builtin_fma(a * mm[i], Un, __builtin_fma(ee[i], W, dd[i]));",
     "const double U = __builtin_fma(a * mm[i], Un, dd[i]);"),
    ("zonal_coefficient_linear_in_dlambda", "a = theta * h->p.D / (mm * (dl * dl));", "a = theta * h->p.D / (mm * dl);", "ebm_runtime.hip"),
    ("annual_means_all_from_the_first_variable", "const size_t base = (size_t)blockIdx.y * (size_t)var_stride + (size_t)col * (size_t)threads * cells;",
     "const size_t base = (size_t)col * (size_t)threads * cells;"),
    ("ring_pieces_reuse_the_first_two_offsets", "const size_t r0 = i * rows_per;", "const size_t r0 = (i % 2) * rows_per;", "ebm_hostcopy.h"),
    # (added after the zonal sweep was partitioned along the circle)
    ("zonal_segment_end_takes_the_wrong_neighbour", "auto rhs = [&](int s_) { return __builtin_fma(a, su[o + (size_t)((s_ + 1) % S) * P], sg[o + (size_t)s_ * P]); };",
     "auto rhs = [&](int s_) { return __builtin_fma(a, su[o + (size_t)s_ * P], sg[o + (size_t)s_ * P]); };"),
    ("zonal_reduced_diagonal_without_the_spike_sum", "const double a2 = a * ep_last, B2 = B - a * cp_last - a * alpha;", "const double a2 = a * ep_last, B2 = B - a * cp_last;", "ebm_runtime.hip"),
    ("integrate_restarts_model_time", "f, diag, clock0 + tinx - 1,", "f, diag, tinx - 1,", "ebm_runtime.hip"),
    ("as_of_query_inverted", "    if (have != step)", "    if (have == step)", "ebm_runtime.hip"),
    # seventh batch: the LDS-resident fused-K kernel, its compact solve, its halo exchange and its vote
    ("resident_diagnostics_of_the_first_step", "const bool diag = a.write_diag != 0 && step + 1 == nloop;", "const bool diag = a.write_diag != 0 && step == 0;"),
    ("resident_newton_stops_after_one_iteration", "} while (it < kMaxNewton && again);", "} while (false);"),
    ("resident_extension_matrix_diagonal_sign", "rb[i] = 1.0 + p.theta_imex * (qlo[i] + qup[i]);", "rb[i] = 1.0 - p.theta_imex * (qlo[i] + qup[i]);"),
    ("resident_floe_size_not_written_back", "            sD(i) = valid ? o.q[Q_D] : 0.0;\n", ""),
    ("resident_vote_reads_only_the_first_wave", "for (int w = 0; w < TT / 64; ++w) any |= F[w];", "for (int w = 0; w < 1; ++w) any |= F[w];"),
    ("wave_halo_takes_its_own_edge", "const double pl = E[w > 0 ? w - 1 : 0],", "const double pl = E[w],"),
    # (two buffer-aliasing mutants of the compact solve were tried and are NOT in this list — the reduction's second buffer
    # overlapping the first by a third, the interface solution written into the buffer the reduction is still read from: both
    # are RACES between waves that resume from the same barrier within a few cycles of each other, both passed every test on
    # the box, and no test can catch such a race deterministically (no GPU sanitizer on this pool).  The buffer plan is argued
    # in the comments of partition_solve_r instead.  What a wrong buffer does deterministically is covered by the next one.)
    ("compact_summaries_overrun_into_the_state", "double *const W1 = COMPACT ? P0 : P1;", "double *const W1 = P1;"),
    ("resident_state_words_one_slot_low", "win[((4 + 4 * (F) + (i)) * T) >> 13][((4 + 4 * (F) + (i)) * T) & 8191]", "win[((3 + 4 * (F) + (i)) * T) >> 13][((3 + 4 * (F) + (i)) * T) & 8191]"),
    # eighth batch: ebm_integrate's fused stretches
    ("resident_sums_of_the_second_pair_land_on_the_first", "(unsigned)((i / 2) * 2 * T) + 2u * (unsigned)td,", "2u * (unsigned)td,"),
    ("integrate_fused_stretch_restarts_its_forcing", "(int)n, f_steps ? f_steps + (tinx - 1) : nullptr, 0, h->integrate_spl,", "(int)n, f_steps, 0, h->integrate_spl,", "ebm_runtime.hip"),
    ("integrate_fused_stretch_one_table_entry_late", "rc = fused_range(h, tinx - 1, clock0 + tinx - 1,", "rc = fused_range(h, tinx, clock0 + tinx - 1,", "ebm_runtime.hip"),
    ("integrate_fuses_through_the_winter_snapshot", "if ((ti_ == winter_inx && (winter || hm_winter)) || (ti_ == summer_inx && (summer || hm_summer))) return false;",
     "if (ti_ == summer_inx && (summer || hm_summer)) return false;", "ebm_runtime.hip"),
]


def build(only=None):
    src = os.path.join(ROOT, "energybalancemodel.jl_amd", "csrc")
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    for name, old, new, *where in MUTANTS:
        if only and name not in only:
            continue
        work = f"/tmp/mutant_{name}"
        shutil.rmtree(work, ignore_errors=True)
        os.makedirs(os.path.join(work, "energybalancemodel.jl_amd"))
        shutil.copytree(src, os.path.join(work, "energybalancemodel.jl_amd", "csrc"), ignore=shutil.ignore_patterns("build"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(work, "include"))
        path = os.path.join(work, "energybalancemodel.jl_amd", "csrc", where[0] if where else "ebm_kernels.hip")
        text = open(path).read()
        assert text.count(old) == 1, (name, text.count(old))
        open(path, "w").write(text.replace(old, new))
        out = os.path.join(ROOT, "build", f"libebm_mut_{name}.so")
        subprocess.check_call(["make", "-j8", "-C", os.path.dirname(path), f"OUT={out}", f"BUILD={work}/obj"], stdout=subprocess.DEVNULL)
        print("built", out, flush=True)


if __name__ == "__main__":
    if sys.argv[1:2] == ["build"]:
        build(sys.argv[2:])
    else:
        print("\n".join(m[0] for m in MUTANTS))
