"""C++ host mirror (include/rs_tfhe_hip.hpp): compiles on CPU; its reference-style test
program (tests/cpp/test_mirror.cpp) runs on the GPU through the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def test_cpp_mirror_builds():
    subprocess.check_call(["make", "-C", CPP])
    assert os.path.exists(os.path.join(CPP, "build", "test_mirror"))


def test_device_fft_algebra_on_host():
    """fft512.hpp's passes / transposes / twiddle table, compiled for the CPU and driven in lock-step over
    64 emulated lanes, equal the definition of KlemsaProcessor::ifft / fft (klemsa.rs:88-150) to 1e-14
    relative, and the stage API's rounding is f64::round on exact ties (klemsa.rs:145-146)."""
    subprocess.check_call(["make", "-C", CPP, "build/test_fft_host"])
    r = subprocess.run([os.path.join(CPP, "build", "test_fft_host")], capture_output=True, text=True, timeout=120)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "all checks passed" in r.stdout


def test_front_end_queue_under_thread_sanitizer():
    """rs-tfhe_amd/csrc/combine_queue.hpp (lock-free arrival list, lane bits, one futex word, lingering leader) compiled
    with -fsanitize=thread and driven by up to 64 threads with a stand-in for the launch: every request served exactly
    once with its own result, no report from ThreadSanitizer, quiesce / with_idle_lanes beside the traffic."""
    subprocess.check_call(["make", "-C", CPP, "build/test_combine_queue"])
    r = subprocess.run([os.path.join(CPP, "build", "test_combine_queue")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 second_deadlock_stack=1"))
    print(r.stdout[-3000:], r.stderr[-6000:])
    assert r.returncode == 0 and "all queue checks passed" in r.stdout
    assert "ThreadSanitizer" not in r.stderr


@pytest.mark.gpu
def test_cpp_mirror_runs_on_gpu():
    exe = os.path.join(CPP, "build", "test_mirror")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", CPP])
    # the oracle's key generation and the host-side checks use OpenMP: keep the team within the container's CPU quota
    # (a 256-thread team on 16 CPUs' worth of time tripled the run)
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "16"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-2000:]
    assert "all C++ mirror tests passed" in r.stdout
