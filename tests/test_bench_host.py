"""bench.py's host-side helpers on the CPU (the GPU legs run on the MI355X box): core counting under a cgroup quota,
the algorithmic-byte formula of SURVEY.md section 8d, the BASELINE config naming, the parity statistics."""
import builtins
import io
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_match_survey_8d():
    assert bench.algorithmic_bytes(25, 13, 3, False, 8) == 8536 and bench.algorithmic_bytes(25, 13, 3, False, 4) == 4268
    assert bench.algorithmic_bytes(25, 7, 3, False, 8) == 7336
    assert bench.algorithmic_bytes(25, 12, 2, True, 4) == 4160 and bench.algorithmic_bytes(25, 12, 2, True, 8) == 8320


def test_baseline_config_names():
    assert bench.baseline_config_of("k13", 65536, 1, "f64").startswith("BASELINE configs[2]")
    assert bench.baseline_config_of("k13", 32768, 8, "f64") == "BASELINE configs[3]"
    assert bench.baseline_config_of("k13", 4096, 1, "f64") == "BASELINE configs[1]"
    assert bench.baseline_config_of("k12_admit", 65536, 1, "mixed") == "BASELINE configs[4]"
    assert bench.baseline_config_of("k7", 100, 1, "f64") == "no BASELINE config"


def _fake_open(files):
    real = builtins.open

    def opener(path, *a, **k):
        if path in files:
            if files[path] is None:
                raise OSError(path)
            return io.StringIO(files[path])
        if str(path).startswith("/sys/fs/cgroup"):
            raise OSError(path)
        return real(path, *a, **k)
    return opener


def test_effective_cores_honours_affinity_and_cgroup_quota(monkeypatch):
    """The GPU box shows 256 hardware threads, an affinity mask of 256 and a cgroup v2 quota of 16 cores: 16 workers, not 256."""
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)))
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    assert bench.effective_cores() == (16, 256, 16.0)
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "max 100000\n"}))
    assert bench.effective_cores() == (256, 256, None)
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": None, "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "350000\n",
                                                      "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert bench.effective_cores() == (3, 256, 3.5)                         # cgroup v1, fractional quota rounds down
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: {0, 1})
    monkeypatch.setattr(builtins, "open", _fake_open({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    assert bench.effective_cores() == (2, 2, 16.0)                          # the mask is the tighter bound


def test_parity_statistics():
    rng = np.random.default_rng(0)
    ref = rng.normal(size=(50, 25))
    u = ref * (1 + 1e-9)
    u[7] += 1e-3 * np.abs(ref[7]).max()
    r = bench._parity_plain(u, ref, 1e-5, "note")
    assert r["n"] == 50 and r["n_over_tol"] == 1 and 0.9e-3 < r["max_rel_err"] < 1.1e-3 and r["median_rel_err"] < 2e-9


def test_cpu_legs_run_without_a_gpu():
    """The CPU baseline (forked workers behind a barrier) and the chained-oracle reference of the from_q leg need no GPU:
    small batch, short windows."""
    from irl_control_amd import synth
    from irl_control_amd.rigid_body import RigidBodyModel
    lay, gains, arr = synth.make_batch("k13", 1200, seed=1)
    cb, ref, idx = bench.cpu_baseline(lay, gains, arr, seconds_single=0.3, seconds_all=0.5)
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and 0 < cb["parallel_efficiency"] <= 1.5
    assert cb["single_core"]["cores"] == 1 and "vectorised" in cb and len(idx) >= 512 and np.all(np.isfinite(ref[list(idx)]))
    assert "affinity mask" in cb["sample"] and "parallel efficiency" in cb["sample"]
    model = RigidBodyModel.load("dual_ur5")
    q, qd = model.random_state(np.random.default_rng(2), 64)
    u = bench.from_q_reference(lay, gains, q, qd, arr["tgt_pose"][:64], 32, 2)
    assert u.shape == (32, 25) and np.all(np.isfinite(u))
    assert bench.oracle_reference(lay, gains, arr, 10, 14).shape == (4, 25)


def test_train_summary_reconciles_span_period_and_events():
    """irlosc_time_trains rows -> the figures of roofline.untraced: span = end - start per train, period = start to start."""
    tt = np.array([[1.00, 0.0, 900.0, 1600.0], [1.00, 850.0, 1760.0, 1610.0], [1.02, 1700.0, 2590.0, 1590.0], [0.99, 2550.0, 3445.0, 1600.0]])
    s = bench.train_summary(tt, 8)
    assert s["trains"] == 4 and s["steps_per_train"] == 8
    assert s["period_us"]["median"] == 850.0 and 895.0 <= s["kernel_span_us"]["median"] <= 910.0
    assert 45.0 <= s["overlap_us_median"] <= 60.0 and 990.0 <= s["event_pair_us"]["median"] <= 1010.0 and s["sclk_mhz"]["median"] == 1600.0


def test_evidence_scalars_lead_the_line():
    """The driver's record of the bench line keeps the first 24 keys of `config` and of `roofline` (from_q_value and
    frac_rocprof were appended behind the cap and lost).  Whatever order the legs fill the dicts in, the evidence scalars lead."""
    assert len(bench.CONFIG_FIRST) <= 24 and len(bench.ROOFLINE_FIRST) <= 24
    prose = {"sharding": "x", "records": "float64", "arithmetic": "f64", "preroll_steps": 1000, "records_from": "physical", "admittance": False,
             "ndev": 3, "steps_per_launch": 8, "slices_per_rank": 1, "sustained_steps": 1, "sustained_ms_per_step": 0.1, "parity_tolerance": 1e-5,
             "from_q_ms_per_step": 0.1, "secondary_mixed_parity_n_over_tol": 0, "synthetic_dense_roofline_frac": 0.5}
    cfg = dict(prose)
    cfg.update({k: 1 for k in bench.CONFIG_FIRST})                       # filled LAST, as bench.py's flat copies are
    got = list(bench.ordered_first(cfg, bench.CONFIG_FIRST))
    assert got[:len(bench.CONFIG_FIRST)] == bench.CONFIG_FIRST and set(got) == set(cfg)
    for must in ("workload", "from_q_value", "from_q_roofline_frac_fp64_valu", "secondary_mixed_value", "secondary_mixed_roofline_frac",
                 "synthetic_dense_value", "end_to_end_host_arrays_value", "parity_n_outside_domain", "sustained_value", "rccl_ranks"):
        assert got.index(must) < 24, must
    roof = {"untraced": {}, "committed_profile": {}, "note": "x"}
    roof.update({k: 1 for k in bench.ROOFLINE_FIRST})
    got = list(bench.ordered_first(roof, bench.ROOFLINE_FIRST))
    for must in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_rocprof", "untraced_kernel_span_us", "untraced_period_us"):
        assert got.index(must) < 24, must
    assert got.index("untraced") >= len(bench.ROOFLINE_FIRST)            # the nested dicts trail
