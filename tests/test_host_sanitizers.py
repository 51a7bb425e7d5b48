"""Host code of the library (BVH builder and flatteners, host half of the scene upload, scene generators, OBJ reader)
under ASan + UBSan on the CPU, including injected allocation failures (every extern "C" entry point must return
RT_E_NOMEM instead of letting std::bad_alloc cross the ABI).
GPU sanitizers are not available on the pool; the kernels are covered by the parity tests instead."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracer_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_sources_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_san"
    srcs = [os.path.join(ROOT, "tests", "host_san", "driver.cpp")] + [os.path.join(CSRC, f) for f in ("bvh.cpp", "scene_host.cpp", "scene_gen.cpp", "obj_loader.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", *srcs, "-o", str(exe), "-lpthread"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)                      # ASan must be the first library in the process
    scratch = tmp_path / "obj"
    scratch.mkdir()
    r = subprocess.run([str(exe), str(scratch)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "0 failures" in r.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_build_under_thread_sanitizer(tmp_path):
    """The parallel host build (fork-join passes of the big nodes, subtree task pool, parallel flatten and record loops) under
    ThreadSanitizer: 700 k triangles take the data-parallel phase, 90 k the task pool alone."""
    exe = tmp_path / "host_tsan"
    srcs = [os.path.join(ROOT, "tests", "host_san", "tsan_driver.cpp")] + [os.path.join(CSRC, f) for f in ("scene_host.cpp", "bvh.cpp", "scene_gen.cpp")]
    b = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", *srcs, "-o", str(exe), "-lpthread"],
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, (r.stdout[-1000:], r.stderr[-4000:])
