"""The host side of librtd.so under the address and undefined-behaviour sanitizers, on the CPU (SURVEY section 5 lists a sanitizer
build of the host side; the GPU pool offers no device sanitizer, and until round 6 the hand-managed pointers of csrc/rtd_api.hip --
plan arenas, window offsets, hand-off slots, retained / lean forms, pinned slabs, the pool -- were exercised on the GPU only).

csrc/rtd_api.hip is compiled AS IT IS with g++ -fsanitize=address,undefined against a stand-in <hip/hip_runtime.h>
(tests/cpu/fake_hip: "device" memory is heap memory, streams and events are tokens, the translation unit's own __global__ kernels
are run thread by thread) and linked with shadow launchers for the kernels of the other translation units, which touch exactly
the extents the real kernels read and write (tests/cpu/host_asan_shadow.cpp).  The REAL Python front end then drives it through
~80 scenarios (tests/cpu/host_asan_driver.py).  What this catches: an access outside a separately allocated buffer (evaluation
buffers, Fourier-mode buffer, gathered arrays, chunk lists, NT tables, staging slabs, temporaries), a window or retained-column
offset that leaves the plan's arena, use after free through the pool, leaks of plans, signed overflow in the size arithmetic.  What
it cannot see: an overlap BETWEEN two buffers carved from one arena (the arena is one allocation; the library copies host images
across adjacent carves on purpose) -- those are held by the GPU tests that compare windowed, retained and one-window plans bit for
bit."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU = os.path.join(ROOT, "tests", "cpu")
LIB = os.path.join(CPU, "librtd_host_asan.so")
STUB_CPU = os.path.join(CPU, "librccl_stub_cpu.so")


def _build():
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    asan = subprocess.run([gxx, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan")
    srcs = [os.path.join(ROOT, "pythonic-disort_amd", "csrc", "rtd_api.hip"), os.path.join(CPU, "host_asan_shadow.cpp")]
    deps = srcs + [os.path.join(CPU, "fake_hip", "hip", "hip_runtime.h"), os.path.join(ROOT, "pythonic-disort_amd", "csrc", "rtd_device.h"),
                   os.path.join(ROOT, "include", "rtd.h")]
    flags = ["-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fPIC", "-shared",
             "-Wno-unused-result", "-I", os.path.join(CPU, "fake_hip")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.run([gxx, *flags, "-Wl,-soname,librtd_host_asan.so", "-x", "c++", *srcs, "-o", LIB, "-ldl"], check=True)
    # the tests' stand-in transport for the CPU: the same source as on the GPU, built against the stand-in runtime (its IPC probe
    # fails there, so it stages through shared memory) and linked to the library above for the fake allocator
    stub_src = os.path.join(ROOT, "tests", "stub", "rccl_stub.cpp")
    if not os.path.exists(STUB_CPU) or os.path.getmtime(STUB_CPU) < max(os.path.getmtime(stub_src), os.path.getmtime(LIB)):
        subprocess.run([gxx, *flags, "-x", "c++", stub_src, "-x", "none", "-L" + CPU, "-l:librtd_host_asan.so", "-Wl,-rpath,$ORIGIN",
                        "-o", STUB_CPU, "-lrt"], check=True)
    return asan


def _drive(asan, **env_over):
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", RTD_LIB=LIB, FAKE_HIP_TOTAL=str(8 << 30))
    for k in ("RTD_POOL_BYTES", "RTD_WORK_BYTES", "RTD_NO_PIPELINE", "RTD_RCCL_STUB"):
        env.pop(k, None)
    env.update(env_over)
    return subprocess.run([sys.executable, os.path.join(CPU, "host_asan_driver.py")], env=env, capture_output=True, text=True, timeout=900)


def test_host_side_of_librtd_is_clean_under_address_and_ub_sanitizers():
    asan = _build()
    r = _drive(asan)
    assert r.returncode == 0 and "ALL SCENARIOS PASSED" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") >= 70


def test_the_harness_sees_an_overrun_of_a_window():
    """Sensitivity: the shadow evaluation kernel made to write past the end of its window's u (what a wrong window offset or a
    short evaluation buffer would do) ends the run with an AddressSanitizer report."""
    asan = _build()
    r = _drive(asan, FAKE_HIP_FAULT="eval_u_overrun")
    assert r.returncode != 0 and "AddressSanitizer" in r.stderr and "heap-buffer-overflow" in r.stderr, r.stderr[-2000:]


@pytest.mark.parametrize("mode,world", [("allgather", 2), ("allgather", 3), ("root", 3), ("layers", 2)])
def test_n_rank_data_plane_on_the_cpu_under_the_sanitizers(mode, world, tmp_path):
    """world-size-2 / 3 on the CPU (the contract's "cover the N > 1 path with world_size-2 tests on CPU"), DATA plane included:
    rank processes run tests/dist_worker.py -- the worker of the GPU tests -- on the sanitizer build of librtd's host code, with the
    tests' stand-in transport built for the CPU (shared-memory staging).  rtd_comm_* of rank > 0 -- slot offsets, the root's
    receive loop, the layer stitch -- run under ASan, and because the shadow evaluation writes values that carry the column's
    identity, a slot that holds the wrong rank's or the wrong column's results fails the worker's bit-for-bit comparison."""
    import json
    asan = _build()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               RTD_LIB=LIB, RTD_RCCL_STUB=STUB_CPU, RCCL_STUB_TIMEOUT_S="120", FAKE_HIP_TOTAL=str(8 << 30))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "RTD_POOL_BYTES", "RTD_WORK_BYTES", "RTD_NO_PIPELINE"):
        env.pop(k, None)
    logs = [open(tmp_path / f"rank_{r}.log", "w+") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), "--mode", mode, "--rank", str(r), "--world", str(world),
                               "--dir", str(tmp_path), "--device", "0"], env=env, stdout=logs[r], stderr=subprocess.STDOUT) for r in range(world)]
    try:
        for p in procs:
            p.wait(timeout=600)
    finally:
        for p in procs:  # exact PIDs only
            if p.poll() is None:
                p.kill()
                p.wait()
    texts = []
    for f in logs:
        f.seek(0)
        texts.append(f.read())
        f.close()
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, texts[r][-3000:])
        assert "AddressSanitizer" not in texts[r] and "runtime error" not in texts[r], texts[r][-3000:]
        res = json.load(open(tmp_path / f"result_{r}.json"))
        assert res["ok"] and res["checks"]["transport"] == "stub:shm", res
        if mode == "allgather":
            assert res["checks"]["q32"]["bit_equal"] and res["checks"]["q8"]["slice_bit_equal"]
        if mode == "root":
            assert res["checks"]["q32"]["bit_equal"] if r == 0 else res["checks"]["q32"]["non_root_fetch_refused"]
