"""CPU-only hygiene: the oracle and the host-side state machines built with AddressSanitizer + UBSan run clean on a
small workload (GPU sanitizers are not available on the pool)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_host_harness_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / 'san_harness')
    src = os.path.join(ROOT, 'tests', 'host', 'san_harness.cpp')
    san = ['-O1', '-g', '-fsanitize=address,undefined', '-fno-omit-frame-pointer']
    objs = []
    for cfile in ('bdrt_oracle.c', 'nuts_oracle.c'):
        o = str(tmp_path / (cfile + '.o'))
        subprocess.check_call(['gcc', '-std=c11'] + san + ['-c', os.path.join(ROOT, 'oracle', cfile), '-o', o])
        objs.append(o)
    subprocess.check_call(['g++', '-std=c++17'] + san + [src] + objs + ['-lm', '-o', exe])
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1', UBSAN_OPTIONS='halt_on_error=1')
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'SAN_OK' in out.stdout
