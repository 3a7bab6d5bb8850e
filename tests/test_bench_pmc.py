"""CPU: bench.py's reader of rocprofv3's counter CSV (the in-run PMC passes behind `roofline.traffic`):
which rows belong to which phi pass, what the calibration kernel is, and what `--lean` implies."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

HEAD = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
        '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value",'
        '"Start_Timestamp","End_Timestamp"\n')


def _row(k, name, ctr, val):
    return f'{k},{k},"Agent 2",1,388,388,1024,37,"{name}",256,0,0,80,0,64,"{ctr}",{val},1,2\n'


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules["bench_module"] = m
    spec.loader.exec_module(m)
    return m


def test_counter_rows_are_attributed_to_the_right_pass(tmp_path):
    b = _bench()
    d = tmp_path / "fetch" / "sub"
    d.mkdir(parents=True)
    item = "void hpf::phi_pass_packed_kernel<hpf::codec_p59, 8, 6, 1>(hpf::PhiArgs)"
    user = "void hpf::phi_pass_packed_kernel<hpf::codec_p59, 8, 6, 0>(hpf::PhiArgs)"
    plain = "void hpf::phi_pass_kernel<double, 8, 7, 2, 1>(hpf::PhiArgs)"
    rows = [_row(1, "void at::native::vectorized_elementwise_kernel<4, foo>(int, foo)", "FETCH_SIZE", 9.0),
            _row(2, item, "FETCH_SIZE", 100.0), _row(3, user, "FETCH_SIZE", 50.0),
            _row(4, item, "FETCH_SIZE", 110.0), _row(5, user, "FETCH_SIZE", 60.0),
            _row(6, plain, "FETCH_SIZE", 7.0),
            _row(7, "hpf::materialize_es_kernel(double*, double*, double const*, double const*, unsigned int)", "FETCH_SIZE", 4000.0),
            _row(8, "hpf::materialize_es_kernel(double*, double*, double const*, double const*, unsigned int)", "FETCH_SIZE", 400.0),
            _row(9, item, "WRITE_SIZE", 1.0)]
    (d / "p_counter_collection.csv").write_text(HEAD + "".join(rows))
    vals, cal = b.parse_counter_csvs(str(tmp_path / "fetch"), "FETCH_SIZE")
    assert vals == {1: [100.0, 110.0, 7.0], 0: [50.0, 60.0]} and cal == [4000.0, 400.0]
    vals, cal = b.parse_counter_csvs(str(tmp_path / "fetch"), "WRITE_SIZE")
    assert vals == {1: [1.0], 0: []} and cal == []


def test_lean_is_what_the_profiled_child_runs():
    """`--lean` must switch every side block off: the child that rocprofv3 profiles may not spawn rocprofv3 itself"""
    src = (ROOT / "bench.py").read_text()
    assert "args.no_pmc = args.no_other_configs = args.no_cpu_baseline = True" in src
    assert '"--lean", "--steps", str(steps), "--warmup", "1"' in src


def test_inner_profiler_runs_do_not_inherit_an_outer_one():
    """bench.py under `rocprofv3 ... -- python bench.py` spawns rocprofv3 passes of its own: they must not inherit the
    outer tool's settings or its preloaded libraries"""
    import bench
    env = {"PATH": "/usr/bin", "TMPDIR": "/tmp", "ROCPROF_OUTPUT_PATH": "/x", "ROCPROF_KERNEL_TRACE": "1",
           "ROCP_TOOL_LIBRARIES": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so", "ROCPROFILER_LIBRARY_CTOR": "1",
           "LD_PRELOAD": "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so:/usr/lib/libjemalloc.so:/opt/rocm/lib/librocprofiler-sdk.so",
           "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    out = bench.clean_profiler_env(env)
    assert out == {"PATH": "/usr/bin", "TMPDIR": "/tmp", "LD_PRELOAD": "/usr/lib/libjemalloc.so", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    env["LD_PRELOAD"] = "/opt/rocm/lib/librocprofiler-sdk.so"
    assert "LD_PRELOAD" not in bench.clean_profiler_env(env)
    assert bench.clean_profiler_env({"A": "1"}) == {"A": "1"}


def test_the_profiled_child_of_a_sharded_run_is_rank_0s_shard():
    """N > 1: the child that rocprofv3 profiles is ONE rank that generates exactly rank 0's user range of
    the same matrix and drives the iteration through the calls a rank makes; N = 1 keeps the plain workload"""
    import argparse
    import bench
    base = dict(n=0, m=0, nnz=0, K=0, scale=1.0, w48=False, w32=False, split_iteration=False)
    a = argparse.Namespace(**base)
    assert bench.pmc_child_workload(a, "C2", None, False) == ["--config", "C2"]
    assert bench.pmc_child_workload(a, "C3", (0, 1249876), True) == ["--config", "C3", "--user-range", "0", "1249876", "--split-iteration"]
    a = argparse.Namespace(**dict(base, scale=0.005, K=50, w48=True))
    assert bench.pmc_child_workload(a, "C3", (10, 20), True) == ["--config", "C3", "--K", "50", "--scale", "0.005", "--w48",
                                                                  "--user-range", "10", "20", "--split-iteration"]
    # HPF_BENCH_FORCE_DIST on one rank: no range, but the split calls
    assert bench.pmc_child_workload(argparse.Namespace(**base), "C2", None, True) == ["--config", "C2", "--split-iteration"]
    src = (ROOT / "bench.py").read_text()
    assert 'raise SystemExit("--user-range / --split-iteration are for a single rank")' in src
    # the child lands on the GPU of the rank that spawned it and carries none of the launcher's rendezvous variables
    assert 'env["LOCAL_RANK"] = str(local_rank)' in src and '"MASTER_ADDR", "MASTER_PORT"' in src
