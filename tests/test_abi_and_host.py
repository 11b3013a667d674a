"""CPU-side checks of the product: the C-ABI library builds for gfx950, loads, exports every symbol
include/slk.h declares, fails loudly without a GPU (no CPU fallback), and the multi-rank bench path
rendezvouses and reduces correctly under gloo with world_size 2.  No compute calls are made here."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def slk():
    import __graft_entry__ as ge
    ge.build()
    from slkpkg import slk as mod
    return mod


def header_functions():
    src = open(os.path.join(ROOT, "include", "slk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(slk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(slk):
    lib = slk.load_library()
    declared = header_functions()
    assert len(declared) >= 25
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(slk.EXPORTS) == declared


def test_layout_queries_do_not_need_a_gpu(slk):
    lib = slk.load_library()
    assert lib.slk_device_count() >= 0
    assert lib.slk_dof(None) == slk.E_INVALID and lib.slk_batch(None) == slk.E_INVALID


def test_no_cpu_fallback(slk):
    if slk.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(slk.SlkError):
        slk.Msckf(np.zeros((1, 13)), np.eye(12))
    import ctypes as C
    h = C.c_void_p()
    cfg = slk.Config(slk.MSCKF, 1, 0, 0, 0, 0, None)
    assert slk.load_library().slk_create(C.byref(cfg), C.byref(h)) == slk.E_NO_DEVICE
    assert slk.load_library().slk_selftest_mfma(0) == slk.E_NO_DEVICE


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "slam-localization_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "oracle" not in txt.lower() or fn == "slk.py" and "oracle" not in txt, (fn,)
    assert "oracle" not in open(os.path.join(ROOT, "include", "slk.h")).read().lower()


def test_cpp_header_facade_compiles_and_links(slk):
    # include/localization/filters/{Usckf,Msckf,State,MtkWrap}.hpp against libslk_hip.so: the reference's
    # class/enum/method names must be usable the way its unit tests use them (no run here: needs a GPU)
    import facade_build
    exe = facade_build.build()
    assert os.path.exists(exe)
    # ... and a caller whose matrices / vectors are fixed-size types of its own (tests/cpp/foreign_matrix.cpp)
    assert os.path.exists(facade_build.build("foreign_matrix"))
    hdr = open(os.path.join(ROOT, "include", "localization", "filters", "Usckf.hpp")).read()
    for name in ("enum CloningMode", "STATEK_I = 3", "class Usckf", "void cloning(int mode)", "setMeasurement(CloningMode mode",
                 "PkAugmentedState() const", "muSingleState(int state = STATEK_I)"):
        assert name in hdr, name
    hdr = open(os.path.join(ROOT, "include", "localization", "filters", "Msckf.hpp")).read()
    for name in ("class Msckf", "unsigned int update(", "getPkSingleState()", "setPk(", "muState() const"):
        assert name in hdr, name


def test_algorithmic_work_figures_match_survey():
    sys.path.insert(0, ROOT)
    import bench
    assert round(bench.algorithmic_flops(60, 8)) == 1294053          # SURVEY.md 8(d), BASELINE.md section 4
    assert round(bench.algorithmic_flops(12, 3)) == 42221
    assert round(bench.algorithmic_flops(198, 8)) == 38742147
    assert round(bench.algorithmic_flops(48, 3, msckf=False)) == 145985      # cfg1, Usckf
    assert bench.algorithmic_bytes(48, 51, 3) == 39032
    assert bench.algorithmic_bytes(60, 69, 8) == 60536
    assert bench.algorithmic_bytes(12, 13, 3) == 3864


def _one_json_line(out):
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def _check_two_rank_line(j):
    assert j["n_gpus"] == 2 and j["dry_run"] is True
    assert j["shard_seeds"] == [0x5EED0000, 0x5EED0001]
    assert j["config"]["global_batch"] == 2 * j["config"]["batch_per_gpu"]
    # max over ranks: rank 1 sleeps 2 ms per step, rank 0 only 1 ms
    assert j["ms_per_step"] >= 2.0
    assert [r["steps"] for r in j["per_rank"]] == [20, 20]
    assert j["per_rank"][1]["elapsed_ns"] >= 20 * 2_000_000
    # the rehearsal batch of rank r reports r filters with a numerical status and 2r all-rejected ones: sums over ranks
    assert j["filters_with_numerical_status"] == 1 and j["filters_all_rejected"] == 2 and j["status_or"] == 9


def test_bench_gpus_flag_starts_the_ranks_itself():
    # the driver's shape of the command WITHOUT a launcher: plain `python bench.py --gpus 2` must start two ranks
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "1", "--dry-run"]
    j = _one_json_line(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300))
    _check_two_rank_line(j)


def test_bench_parent_does_not_touch_torch_or_the_library():
    # the self-launching parent must start its children before anything GPU-related is imported or called
    code = ("import sys, bench\n"
            "bench.subprocess.Popen = lambda *a, **k: (_ for _ in ()).throw(SystemExit(\n"
            "    7 if any(m in sys.modules for m in ('torch', 'slkpkg', 'slk', 'numpy')) else 0))\n"
            "bench.main(['--gpus', '2', '--dry-run'])\n")
    env = {k: v for k, v in os.environ.items() if k != "WORLD_SIZE"}
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stderr[-2000:])


def test_bench_rejects_a_launcher_with_another_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr


def test_bench_failing_rank_fails_the_run():
    # a rank that dies (here: no GPU for the real workload) must take the run down with a non-zero status, not hang
    from slkpkg import slk as mod
    if mod.device_count() > 0:
        pytest.skip("a GPU is visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "no CPU fallback" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_multi_rank_path_under_torchrun():
    # the driver's multi-GPU launch: torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE, bench.py must not re-launch
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "20", "--warmup", "1", "--dry-run"]
    j = _one_json_line(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300))
    _check_two_rank_line(j)


_STUBBED_RANK = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, {root!r})
import bench

class StubBatch:                      # stands where bench.GpuBatch stands: same methods, host arithmetic instead of slk calls
    device, dry = "cpu", False
    def __init__(self, args, rank, local_rank):
        self.rank, self.B, self.N, self.Nq, self.m, self.kind = rank, args.batch, 60, 69, 8, "msckf"
        self.seed = bench.SEED0 + rank
        self.calls = 0
    def step(self):
        self.calls += 1
        time.sleep(0.002 if self.rank == 1 else 0.0005)
    def sync(self): pass
    def timer_start(self): self.t = time.perf_counter(); self.calls_at_start = self.calls
    def timer_stop(self):
        assert self.calls - self.calls_at_start == 7      # exactly K timed steps between the events
        return (time.perf_counter() - self.t) * 1e3
    def status(self):
        st = np.zeros(self.B, dtype=np.int32)
        st[0:2 + self.rank] = 4                            # SLK_ST_SINGULAR on 2 (rank 0) / 3 (rank 1) filters
        st[5] = 8                                          # one all-rejected filter per rank
        return st

args = bench.parse_args(["--gpus", "2", "--steps", "7", "--warmup", "3", "--batch", "64", "--no-cpu-baseline", "--rate-1gpu", "1000"])
bench.run_rank(args, int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), 2, factory=StubBatch)
"""


def test_bench_real_control_path_with_the_filter_batch_stubbed(tmp_path):
    # NOT --dry-run: run_rank / report as they run on hardware (warm-up, barrier, timed loop, all_gather of the per-rank
    # counters, MAX of the times, SUM of the status counts, roofline block), the filter batch replaced at the one seam
    script = tmp_path / "stub_rank.py"
    script.write_text(_STUBBED_RANK.format(root=ROOT))
    port = str(29000 + os.getpid() % 500)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]        # only rank 0 prints the line
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert "dry_run" not in j and j["n_gpus"] == 2 and j["steps"] == 7 and j["warmup"] == 3
    assert j["config"]["global_batch"] == 128 and j["scaling"] == "weak"
    assert j["filters_with_numerical_status"] == 5 and j["filters_all_rejected"] == 2 and j["status_or"] == 12
    assert j["ms_per_step"] >= 2.0                                    # the slow rank sets the time
    assert abs(j["value"] - 128 * 7 / (j["ms_per_step"] * 7e-3)) < 1e-6 * j["value"]
    assert [r["seed"] for r in j["per_rank"]] == [0x5EED0000, 0x5EED0001]
    assert j["roofline"]["bound"] == "mfma" and j["roofline"]["kernel_ms"] >= 2.0
    assert abs(j["weak_scaling_efficiency_vs_given_1gpu_rate"] - j["value"] / 2000.0) < 1e-9
    assert "cpu_baseline" not in j                                    # rank 0 at N = 1 only
