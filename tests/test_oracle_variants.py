"""How far would results move if upstream dbot chose a ledger rule differently?  (CPU only.)

The oracle is unpinned: where SURVEY Appendix A left a rule open the restatement fixed one.  Here the
oracle of record (reference semantics: LAZY) runs a tracked, resampled sequence side by side with the SAME source rebuilt with
one rule changed (oracle/rbsensor_oracle.c "EXPOSURE VARIANTS"), on identical inputs and identical histories (both sides
continue with the rule-of-record parents), and the test prints / records per variant
    max |d ll| / max(1, |ll|)          how far a particle's log-likelihood moves
    parent mismatches / children      how many resampled children would draw another parent
    covered pixels that differ        coverage variants: pixels of the rendered masks that change
This sizes the exposure; it pins nothing.  The table is written to profiles/r05_oracle_variant_exposure.json and quoted in DESIGN.md."""
import json
import os

import numpy as np
import pytest

import oracle_binding as ob
import scenarios as sc
from dbot_ros_amd import filter as flt, synth
from test_gpu_reference_semantics import _poses_around

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(variant, cols, rows, n, n_frames, seed, inf_pixels=0):
    om, cam, P = sc.make_scene(("m1",), cols, rows, max_particles=n)
    ref = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY)
    var = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY, variant=variant)
    threads = sc.usable_threads()
    rng = np.random.default_rng(seed)
    dl = rng.normal(0.0, 0.0025, size=(n, 1, 3))
    da = rng.normal(0.0, 0.02, size=(n, 1, 3))
    idx = [np.zeros(n, np.int32), np.zeros(n, np.int32)]
    logw, ll_prev = [np.zeros(n), np.zeros(n)], [np.zeros(n), np.zeros(n)]
    worst, mism, children, cov_diff, cov_total, nan_particles = 0.0, 0, 0, 0, 0, 0
    ref.reset(threads=threads); var.reset(threads=threads)
    for k in range(n_frames):
        truth = synth.truth_pose(1, frame=k)
        d_ref, d_var = ref.render_depth(truth), var.render_depth(truth)
        cov_diff += int((np.isfinite(d_ref) != np.isfinite(d_var)).sum())
        cov_total += int(np.isfinite(d_ref).sum())
        frame = synth.make_frame(d_ref, rows, cols, rng)
        if inf_pixels:      # a driver that reports "too far" as +inf instead of NaN (none we know of does)
            on = np.flatnonzero(np.isfinite(d_ref))
            frame[rng.choice(on, size=min(inf_pixels, len(on)), replace=False)] = np.inf
        ref.set_observation(frame); var.set_observation(frame)
        dl = 0.8 * dl + rng.normal(0.0, 0.0025, size=(n, 1, 3))
        da = 0.8 * da + rng.normal(0.0, 0.02, size=(n, 1, 3))
        poses = _poses_around(truth, dl, da)
        ll = [ref.loglikes_poses(poses, idx[0], update=True, threads=threads), var.loglikes_poses(poses, idx[1], update=True, threads=threads)]
        bad = ~np.isfinite(ll[1])
        nan_particles += int(bad.sum())
        ok = ~bad
        if ok.any():
            worst = max(worst, float((np.abs(ll[0][ok] - ll[1][ok]) / np.maximum(1.0, np.abs(ll[0][ok]))).max()))
        ll[1] = np.where(bad, ll[0], ll[1])     # (a NaN would end the comparison: count it, carry on with the reference's number)
        w, kl = [], []
        for s in range(2):
            logw[s] += ll[s] - ll_prev[s]
            ll_prev[s] = ll[s]
            w.append(flt.normalized_weights(logw[s]))
            kl.append(flt.kl_to_uniform(w[s]))
        if kl[0] > 2.0:
            u = rng.random(n)
            p0, p1 = flt.multinomial_resample(w[0], u), flt.multinomial_resample(w[1], u)
            mism += int((p0 != p1).sum())
            children += n
            dl, da = dl[p0], da[p0]
            idx = [idx[0][p0].copy(), idx[1][p0].copy()]
            for s in range(2):
                ll_prev[s] = ll_prev[s][p0]
                logw[s] = np.zeros(n)
    ref.close(); var.close()
    return {"max_rel_loglik_diff": worst, "parent_mismatches": mism, "children": children, "covered_pixels_that_differ": cov_diff,
            "covered_pixels": cov_total, "particles_with_nan": nan_particles}


def test_exposure_of_the_unpinned_rules():
    cases = [("640x480, 48 particles, 10 frames", 640, 480, 48, 10), ("80x60, 400 particles, 20 frames", 80, 60, 400, 20)]
    table = {}
    for variant in ob.VARIANTS:
        table[variant] = {}
        for name, cols, rows, n, nf in cases:
            table[variant][name] = _run(variant, cols, rows, n, nf, seed=5, inf_pixels=40 if variant == "inf_evaluated" else 0)
    print("\nvariant            case                                max |d ll|/max(1,|ll|)   parents differing      covered px differing")
    for variant, by_case in table.items():
        for name, r in by_case.items():
            print(f"{variant:18s} {name:36s} {r['max_rel_loglik_diff']:.3e}              {r['parent_mismatches']:5d} / {r['children']:<7d}    "
                  f"{r['covered_pixels_that_differ']:5d} / {r['covered_pixels']:<8d}" + (f"   NaN particles {r['particles_with_nan']}" if r["particles_with_nan"] else ""))
    out = os.path.join(ROOT, "profiles", "r05_oracle_variant_exposure.json")
    json.dump({"what": "the oracle of record (LAZY) against the same source with ONE ledger rule changed, identical inputs and histories "
                       "(tests/test_oracle_variants.py); sizes the exposure of the unpinned restatement, pins nothing",
               "variants": {"cov_topleft": "L1: top-left fill rule instead of the closed triangle", "cov_scanline": "L1: per-row spans from the edges' intersections",
                            "cov_centres": "L1: samples at pixel centres (col + 0.5, row + 0.5) instead of integer coordinates",
                            "round_f64": "L6: binary64 throughout instead of float temporaries", "inf_evaluated": "L4: +-inf observations evaluated instead of skipped (40 +inf readings per frame injected)"},
               "table": table}, open(out, "w"), indent=1)
    # what the table must keep saying for DESIGN.md's statements to hold
    for v in ("cov_topleft", "cov_scanline"):      # equivalent rules up to exact ties / rounding of an intersection: no sample flips
        for r in table[v].values():
            assert r["covered_pixels_that_differ"] == 0 and r["max_rel_loglik_diff"] <= 1e-12 and r["parent_mismatches"] == 0
    for r in table["round_f64"].values():          # float temporaries against binary64: far inside north_star's 1e-5
        assert r["max_rel_loglik_diff"] <= 1e-5
    for r in table["cov_centres"].values():        # a half-pixel shift of the sample grid is NOT inside any tolerance: it would show at once
        assert r["max_rel_loglik_diff"] > 1e-5 and r["covered_pixels_that_differ"] > 0
    for r in table["inf_evaluated"].values():      # an evaluated +inf poisons the sum: there is no finite "other answer"
        assert r["particles_with_nan"] > 0


def test_pixel_centre_sampling_is_a_shift_of_the_principal_point():
    """The one consequential open rule (cov_centres above) needs no rebuild should upstream turn out to use it: the renderer meets pixel
    coordinates only as sample points against the projected triangles, so samples at (col + 0.5, row + 0.5) under K are samples at
    (col, row) under K with the principal point moved by (-0.5, -0.5).  The variant oracle with K against the oracle of record with the
    shifted K: the same coverage (but for samples on an edge to rounding) and the same depths to float rounding; log-likelihoods
    agree to 1e-6 -- against 8 % / 240 % between the conventions themselves.  INTEGRATION.md section 5 states the remedy."""
    import copy
    for cols, rows, n in ((640, 480, 24), (80, 60, 96)):
        om, cam, P = sc.make_scene(("m1",), cols, rows, max_particles=n)
        cam_shift = copy.deepcopy(cam)
        cam_shift.camera_matrix = np.array(cam.camera_matrix, dtype=np.float64).copy()
        cam_shift.camera_matrix[0, 2] -= 0.5
        cam_shift.camera_matrix[1, 2] -= 0.5
        centres = ob.Oracle(om, cam, P, max_particles=n, mode=ob.LAZY, variant="cov_centres")
        shifted = ob.Oracle(om, cam_shift, P, max_particles=n, mode=ob.LAZY)
        threads = sc.usable_threads()
        centres.reset(threads=threads); shifted.reset(threads=threads)
        rng = np.random.default_rng(7)
        flips = covered = 0
        worst_depth = worst_ll = 0.0
        idx = [np.zeros(n, np.int32), np.zeros(n, np.int32)]
        for k in range(6):
            truth = synth.truth_pose(1, frame=k)
            da, db = centres.render_depth(truth), shifted.render_depth(truth)
            fa, fb = np.isfinite(da), np.isfinite(db)
            flips += int((fa != fb).sum()); covered += int(fa.sum())
            both = fa & fb
            worst_depth = max(worst_depth, float((np.abs(da[both] - db[both]) / da[both]).max()))
            frame = synth.make_frame(da, rows, cols, rng)
            centres.set_observation(frame); shifted.set_observation(frame)
            poses = synth.particle_poses(truth, n, rng, scale=1.0)
            la = centres.loglikes_poses(poses, idx[0], update=True, threads=threads)
            lb = shifted.loglikes_poses(poses, idx[1], update=True, threads=threads)
            worst_ll = max(worst_ll, float((np.abs(la - lb) / np.maximum(1.0, np.abs(la))).max()))
            p = np.sort(rng.choice(n, size=n)).astype(np.int32)
            idx = [p.copy(), p.copy()]
        record = ob.Oracle(om, cam, P, max_particles=1, mode=ob.LAZY)      # (not vacuous: under the SAME K the two conventions differ)
        d0 = record.render_depth(synth.truth_pose(1, frame=0))
        assert (np.isfinite(d0) != np.isfinite(centres.render_depth(synth.truth_pose(1, frame=0)))).sum() > 0
        record.close()
        centres.close(); shifted.close()
        print(f"\n{cols}x{rows}: pixel centres under K vs integer samples under K - (0.5, 0.5): {flips} of {covered} covered px flip, "
              f"depth {worst_depth:.1e}, log-likelihood {worst_ll:.1e}")
        assert flips <= max(2, covered // 20000) and worst_depth <= 1e-6 and worst_ll <= 1e-6
