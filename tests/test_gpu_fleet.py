"""Multi-GPU behind the C ABI, on the one GPU the test box has:

* RCCL itself, executed: a world-size-1 communicator inside libhector_amd.so (hx_comm_init_rank ->
  ncclCommInitRank, hx_ensemble_stats -> ncclAllGather on the core's stream), with and without
  PyTorch in the process, and torch.distributed's own nccl group at world size 1 through bench.py;
* the sharded core (hx_newcore_devices) with the device list [0, 0] -- the rehearsal switch, since
  RCCL refuses two ranks on one GPU: every routed verb against ONE core over the same members;
* bench.py --gpus N: refuses a box with fewer GPUs, launches its own ranks when started as plain
  python, reports n_gpus == N and the collective's world size.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import hector_amd
from hector_amd import ensemble
from hector_amd.core import comm_unique_id
from hector_amd.distributed import stats_numpy
from conftest import ROOT, SCENARIO

pytestmark = pytest.mark.gpu

VARS = ["CO2_concentration", "global_tas"]


def _check_stats(st, core, y0, y1):
    for k, v in enumerate(VARS):
        ref = stats_numpy(core.fetchvars(v, (y0, y1)))
        np.testing.assert_array_equal(st[k][:, 0], ref[:, 0])          # count
        np.testing.assert_allclose(st[k][:, 1:3], ref[:, 1:3], rtol=1e-12)
        np.testing.assert_array_equal(st[k][:, 3:], ref[:, 3:])        # min, max


def test_rccl_world_of_one_inside_the_library(hip_lib):
    """ncclCommInitRank + ncclAllGather really execute on the device (world size 1)."""
    n = 1000
    S, q10 = ensemble.ecs_q10(n)
    c = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    c.setvar("S", S).setvar("q10_rh", q10)
    assert c.comm_info()[0] == 0
    c.comm_init_rank(1, 0, comm_unique_id(hip_lib))
    world, first, backend = c.comm_info()
    assert (world, first) == (1, 0) and backend.startswith("rccl "), backend
    c.run(1900)
    st = c.ensemble_stats(VARS, (1745, 1900))
    _check_stats(st, c, 1745, 1900)
    # hx_stats_device of a core with a communicator takes the same path
    import torch
    d = torch.zeros((156, 5), dtype=torch.float64, device="cuda:0")
    c.stats_device("global_tas", 1745, 1900, d.data_ptr())
    np.testing.assert_array_equal(d.cpu().numpy(), st[1])
    with pytest.raises(hector_amd.HectorAmdError, match="already has a communicator"):
        c.comm_init_rank(1, 0, comm_unique_id(hip_lib))
    c.shutdown()


RCCL_WITHOUT_TORCH = """
import sys, numpy as np, hector_amd
from hector_amd.core import comm_unique_id
assert "torch" not in sys.modules
c = hector_amd.Core(hector_amd.DEFAULT_SCENARIO, 256, device=0)
c.comm_init_rank(1, 0, comm_unique_id())
c.run(1800)
st = c.ensemble_stats(["global_tas"], (1745, 1800))
x = c.fetchvars("global_tas", (1745, 1800))
assert (st[0][:, 0] == 256).all() and (st[0][:, 3] == x.min(1)).all() and (st[0][:, 4] == x.max(1)).all()
assert "torch" not in sys.modules
rc = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
assert len(rc) == 1, rc
print("ok", c.comm_info()[2], rc[0])
"""


def test_rccl_without_pytorch_in_the_process():
    """An R / C++ host: no torch anywhere; the library loads the system's librccl.so.1 itself."""
    r = subprocess.run([sys.executable, "-c", RCCL_WITHOUT_TORCH], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.strip().startswith("ok rccl ")


def test_sharded_core_on_a_duplicate_device_list_equals_one_core(hip_lib, monkeypatch):
    monkeypatch.setenv("HECTOR_AMD_FLEET_REHEARSAL", "1")
    n = 5000   # 1667 + 1667 + 1666
    S, q10 = ensemble.ecs_q10(n)
    one = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    many = hector_amd.Core(SCENARIO, n, devices=[0, 0, 0], lib_path=hip_lib)
    assert many.shards() == ([0, 0, 0], [0, 1667, 3334, 5000])
    for c in (one, many):
        c.set_pair_kernel_limit(0)   # the same kernel on both sides: bitwise comparison
        c.setvar("S", S, "degC").setvar("q10_rh", q10)
        c.run(2300, wait=False)
    for v in VARS:
        np.testing.assert_array_equal(one.fetchvars(v, (1745, 2300)), many.fetchvars(v, (1745, 2300)))
    assert (many.status() == 0).all()
    st = many.ensemble_stats(VARS, (1745, 2300))
    _check_stats(st, one, 1745, 2300)
    assert many.comm_info()[0] == 3 and "rehearsal" in many.comm_info()[2]
    assert many.last_run_ms() > 0
    # reset + rerun on every shard
    for c in (one, many):
        c.reset(1745)
        c.setvar("beta", [0.4])
        c.run(1800)
    np.testing.assert_array_equal(one.fetchvars("CO2_concentration", (1745, 1800)),
                                  many.fetchvars("CO2_concentration", (1745, 1800)))
    one.shutdown(); many.shutdown()


def test_eight_shards_on_one_device_equal_one_core(hip_lib, monkeypatch):
    """hx_newcore_devices([0] * 8) under the rehearsal switch: BASELINE configs[3]'s control flow
    behind the C ABI -- eight shards, eight member blocks, eight host threads, the statistics of
    every shard gathered and folded in shard order -- against ONE core over the same members."""
    monkeypatch.setenv("HECTOR_AMD_FLEET_REHEARSAL", "1")
    n = 8 * 2048
    S, q10 = ensemble.ecs_q10(n)
    one = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    many = hector_amd.Core(SCENARIO, n, devices=[0] * 8, lib_path=hip_lib)
    assert many.shards() == ([0] * 8, [2048 * k for k in range(9)])
    for c in (one, many):
        c.set_pair_kernel_limit(0)   # the same kernel on both sides: bitwise comparison
        c.setvar("S", S, "degC").setvar("q10_rh", q10)
        c.run(2300, wait=False)
    for v in VARS:
        np.testing.assert_array_equal(one.fetchvars(v, (1745, 2300)), many.fetchvars(v, (1745, 2300)))
    assert (many.status() == 0).all()
    _check_stats(many.ensemble_stats(VARS, (1745, 2300)), one, 1745, 2300)
    assert many.comm_info()[0] == 8 and "rehearsal" in many.comm_info()[2]
    one.shutdown(); many.shutdown()


def test_duplicate_devices_are_refused_without_the_switch(hip_lib, monkeypatch):
    monkeypatch.delenv("HECTOR_AMD_FLEET_REHEARSAL", raising=False)
    with pytest.raises(hector_amd.HectorAmdError, match="appears twice"):
        hector_amd.Core(SCENARIO, 128, devices=[0, 0], lib_path=hip_lib)


def test_a_device_the_box_does_not_have_is_an_error(hip_lib):
    import torch
    nd = torch.cuda.device_count()
    with pytest.raises(hector_amd.HectorAmdError, match="invalid device"):
        hector_amd.Core(SCENARIO, 128, devices=list(range(nd + 1)), lib_path=hip_lib)


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_PORT"] = str(29900 + os.getpid() % 90)
    env.update(env_extra or {})
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
           "--members", "2048", "--no-cpu-baseline", "--no-other-configs"] + args
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_refuses_more_gpus_than_the_box_has():
    import torch
    nd = torch.cuda.device_count()
    r, out = _bench(["--gpus", str(nd + 1)])
    assert r.returncode != 0 and out is None
    assert "HIP device(s)" in r.stderr
    r, out = _bench(["--gpus", str(nd + 1), "--single-process"])
    assert r.returncode != 0 and out is None


@pytest.mark.parametrize("collective", ["native", "torch"])
def test_bench_runs_the_rccl_collective_in_a_world_of_one(collective):
    """dist.init_process_group("nccl", device_id=...) + the collective of each flavour, on
    hardware, with one rank: the exact code path the 8-GPU run takes."""
    r, out = _bench(["--gpus", "1", "--force-collective", "--collective", collective])
    assert r.returncode == 0, r.stderr[-3000:]
    assert out["n_gpus"] == 1 and out["config"]["collective_world_size"] == 1
    assert "RCCL" in out["config"]["collective_backend"], out["config"]["collective_backend"]
    assert ("libhector_amd.so" in out["config"]["collective_backend"]) == (collective == "native")
    assert out["config"]["members_in_statistics"] == 2048
    assert out["config"]["members_with_model_errors"] == 0


def test_plain_python_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no torch.distributed.run in front): two ranks, n_gpus 2.  On
    this one-GPU box through the gloo rehearsal (two ranks share device 0)."""
    r, out = _bench(["--gpus", "2", "--dist-backend", "gloo"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert out["n_gpus"] == 2 and out["config"]["collective_world_size"] == 2
    assert out["config"]["global_members"] == 4096 and out["config"]["members_in_statistics"] == 4096


def test_single_process_bench_over_a_device_list():
    r, out = _bench(["--gpus", "2", "--single-process", "--dist-backend", "gloo"],
                    {"HECTOR_AMD_FLEET_REHEARSAL": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert out["n_gpus"] == 2 and out["config"]["collective_world_size"] == 2
    assert out["config"]["members_in_statistics"] == 4096
    S, q10 = ensemble.ecs_q10(4096)
    c = hector_amd.Core(SCENARIO, 4096, device=0)
    c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)").run(2300)
    tg = c.fetchvars("global_tas", (2300, 2300))[0]
    assert abs(out["config"]["tgav_2300_mean_K"] - tg.mean()) < 1e-9
    c.shutdown()


def test_c_host_example_on_the_gpu(hip_lib, tmp_path):
    """examples/multi_gpu_host.c against the product library: one GPU listed once -- the
    statistics still go through hx_ensemble_stats -- and, with the rehearsal switch, twice."""
    exe = str(tmp_path / "multi_gpu_host")
    libdir = os.path.dirname(hip_lib)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "multi_gpu_host.c"), "-o", exe,
                           "-L", libdir, "-lhector_amd", "-Wl,-rpath," + libdir, "-lm"])
    env = dict(os.environ)
    for devs, extra in (("0", {}), ("0,0", {"HECTOR_AMD_FLEET_REHEARSAL": "1"})):
        env.update(extra)
        r = subprocess.run([exe, SCENARIO, "4096", devs, "2100"], capture_output=True, text=True,
                           timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "global_tas 2100: members 4096 " in r.stdout and "members with model errors: 0" in r.stdout


def test_sharded_core_over_every_visible_device_equals_one_core(hip_lib):
    """hx_newcore_devices over range(hipGetDeviceCount()): on a one-GPU box this repeats the
    one-device list; on any box with more it is the first execution of the fleet on distinct
    devices -- per-shard host threads, ncclCommInitRank of several ranks inside one group, the
    grouped all-gather, strided fetchvars into one host array -- against ONE core over the same
    members, bit for bit, and the gathered statistics against a host reduction."""
    import torch
    nd = torch.cuda.device_count()
    n = 3000 * nd + 7   # ragged blocks
    S, q10 = ensemble.ecs_q10(n)
    one = hector_amd.Core(SCENARIO, n, device=0, lib_path=hip_lib)
    many = hector_amd.Core(SCENARIO, n, devices=list(range(nd)), lib_path=hip_lib)
    devs, offs = many.shards()
    assert devs == list(range(nd)) and offs[0] == 0 and offs[-1] == n
    for c in (one, many):
        c.set_pair_kernel_limit(0)
        c.setvar("S", S, "degC").setvar("q10_rh", q10)
        c.run(2300, wait=False)
    for v in VARS:
        np.testing.assert_array_equal(one.fetchvars(v, (1745, 2300)), many.fetchvars(v, (1745, 2300)))
    assert (many.status() == 0).all()
    st = many.ensemble_stats(VARS, (1745, 2300))
    _check_stats(st, one, 1745, 2300)
    world, first, backend = many.comm_info()
    if nd > 1:
        assert world == nd and first == 0 and backend.startswith("rccl "), (world, backend)
    one.shutdown(); many.shutdown()


def test_bench_two_ranks_over_rccl_on_two_gpus():
    """One process per GPU, world size 2, the library's own communicator (ncclCommInitRank with
    world > 1) and ONE ncclAllGather per step: needs two devices (skipped on the builder's box,
    run by whoever has them -- the driver's 8-GPU node takes the same path at --gpus 8)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    r, out = _bench(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert out["n_gpus"] == 2 and out["config"]["collective_world_size"] == 2
    assert "RCCL ncclAllGather issued by libhector_amd.so" in out["config"]["collective_backend"]
    assert out["config"]["global_members"] == 4096 and out["config"]["members_in_statistics"] == 4096
    assert out["config"]["native_collective_fallback"] is None
    # the gathered statistics are those of one core over the same 4096 members
    S, q10 = ensemble.ecs_q10(4096)
    c = hector_amd.Core(SCENARIO, 4096, device=0)
    c.setvar("S", S, "degC").setvar("q10_rh", q10, "(unitless)").run(2300)
    tg = c.fetchvars("global_tas", (2300, 2300))[0]
    assert abs(out["config"]["tgav_2300_mean_K"] - tg.mean()) < 1e-9
    c.shutdown()


def test_bench_multi_gpu_default_times_the_named_configurations():
    """`bench.py --gpus N` without --members: the line's value is the N = 1 workload weak-scaled
    (BASELINE configs[2], 65 536 members per GPU: one curve over N) and other_configs[0] is
    BASELINE configs[3]'s shape (131 072 members per GPU, 1 048 576 at N = 8); both are at the
    line's top level (value_per_gpu_workload, first_run_kernel_ms), both checked for complete
    statistics.  Two gloo ranks on this box's one GPU (rehearsal: the line says so)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_PORT"] = str(29800 + os.getpid() % 90)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and "invalid" not in out
    cfg = out["config"]
    assert cfg["members_per_gpu"] == 65536 and cfg["global_members"] == 131072
    assert cfg["members_in_statistics"] == 131072 and cfg["collective_world_size"] == 2
    assert "configs[2]" in cfg["workload"] and cfg["rehearsal_not_rccl"] is True
    assert out["roofline"]["kernel"].startswith("hx_run_kernel<1")
    o = out["other_configs"][0]
    assert o["members_per_gpu"] == 131072 and o["global_members"] == 262144
    assert o["members_in_statistics"] == 262144 and "configs[3]" in o["workload"]
    assert o["kernel"].startswith("hx_run_kernel<HX_B1W2")
    assert "hip_runtime" in cfg["versions"] and "compiler" in cfg["versions"]
    # both workloads at the top level of the line, each with its one-shot (first run) kernel time
    assert out["value_per_gpu_workload"]["65536"] == out["value"]
    assert out["value_per_gpu_workload"]["131072"] == o["value"]
    for k in ("65536", "131072"):
        assert out["first_run_kernel_ms"][k] > 0 and out["kernel_ms"][k] > 0
    assert "65536 members on EVERY GPU" in out["scaling_workload"]
