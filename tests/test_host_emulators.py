"""The workgroup FFT / DCT index arithmetic of the HIP kernels, executed thread by
thread on the CPU (tests/host/*.cpp include the same headers the kernels use)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('name', ['fft_emulator', 'dct_emulator', 'mrfft_emulator'])
def test_emulator(name, tmp_path):
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('g++ not available')
    exe = str(tmp_path / name)
    src = os.path.join(ROOT, 'tests', 'host', name + '.cpp')
    subprocess.run([gxx, '-O2', '-std=c++17', '-I', os.path.join(ROOT, 'pygpa_amd', 'csrc'), src, '-o', exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert out.stdout.strip().endswith('OK')
