"""Host-only product code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; the GPU pool
offers no sanitizers).  Covers filter design, the call planner with pending frames and both entry
points' block rules, the filter-change re-alignment and the phase rescaling on a grid of rate pairs
that includes absurd ones (1 Hz, 4 MHz) and lengths / capacities up to 2^32."""
import os
import shutil
import subprocess

import pytest

from golden_util import ROOT


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "node-speex-resampler_amd", "csrc")
    exe = str(tmp_path / "san_host")
    build = subprocess.run(
        ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
         "-I" + csrc, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "san_host.cpp"),
         os.path.join(csrc, "filter_design.cpp"), os.path.join(csrc, "stream_plan.cpp"),
         os.path.join(csrc, "devices_rule.cpp"), "-o", exe],
        capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0 and "sanitizer run ok" in run.stdout, run.stdout[-1000:] + run.stderr[-3000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_many_call_worker_threads_are_clean_under_thread_sanitizer(tmp_path):
    """the persistent (device, lane) worker threads of the many-states call (csrc/unit_workers.cpp) and
    the live-count placement rule under ThreadSanitizer, in the shape Batch::process_host_many drives them -- several
    callers at once, nested helper jobs, per-stage locks, shutdown and a second life (tools/san_workers.cpp)."""
    csrc = os.path.join(ROOT, "node-speex-resampler_amd", "csrc")
    exe = str(tmp_path / "san_workers")
    build = subprocess.run(
        ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-I" + csrc, "-I" + os.path.join(ROOT, "include"),
         os.path.join(ROOT, "tools", "san_workers.cpp"), os.path.join(csrc, "unit_workers.cpp"),
         os.path.join(csrc, "devices_rule.cpp"), "-o", exe],
        capture_output=True, text=True, timeout=600)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no thread-sanitizer runtime: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    if run.returncode != 0 and "FATAL: ThreadSanitizer: unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert run.returncode == 0 and "sanitizer run ok" in run.stdout and "WARNING: ThreadSanitizer" not in run.stderr, \
        run.stdout[-1000:] + run.stderr[-3000:]
