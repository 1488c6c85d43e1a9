"""The C-ABI library builds, loads without a GPU and exports every symbol include/smx.h declares."""
import ctypes
import os
import re

from common import ROOT


def _declared_symbols(header="smx.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from surfelmeshing_amd import _lib, build
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.SO_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 35
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    # the Python binding lists exactly the header's functions
    assert sorted(_lib.EXPORTS) == declared
    # the native frame driver (include/smx_driver.h) lives in the same library
    from surfelmeshing_amd.pipeline import DRIVER_EXPORTS
    driver = _declared_symbols("smx_driver.h")
    assert len(driver) >= 13 and not [s for s in driver if not hasattr(lib, s)]
    assert sorted(DRIVER_EXPORTS) == driver


def test_integrate_params_layout_matches_oracle_and_header():
    from surfelmeshing_amd._lib import IntegrateParams, BufferDesc, SurfelBuffersCPU
    import oracle
    assert ctypes.sizeof(IntegrateParams) == 40 == ctypes.sizeof(oracle.IntegrateParams)
    assert [f[0] for f in IntegrateParams._fields_] == [f[0] for f in oracle.IntegrateParams._fields_]
    assert ctypes.sizeof(BufferDesc) == 24          # CUDABuffer_<T>: T*, int, int, size_t
    assert ctypes.sizeof(SurfelBuffersCPU) == 16 + 8 * 8


def test_no_device_errors_are_loud():
    """Without a GPU the product path must fail loudly (no CPU fallback)."""
    from surfelmeshing_amd import _lib, api
    import pytest
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.SmxError):
        api.CUDASurfelReconstruction(1000, api.PinholeCamera4f(64, 48, 50.0, 50.0, 32.0, 24.0))
    with pytest.raises(_lib.SmxError):
        api.SurfelNeighborIndex()


def test_headers_are_plain_c_and_pod_sizes_match(tmp_path):
    """include/smx.h and include/smx_driver.h compile as C11 (no C++ / torch types at the boundary); the PODs that
    cross it have the sizes the ctypes mirrors assume."""
    import subprocess
    src = tmp_path / "abi_probe.c"
    src.write_text(
        '#include <stdio.h>\n#include "smx.h"\n#include "smx_driver.h"\n'
        'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(smx_buffer_desc), sizeof(smx_integrate_params),\n'
        '  sizeof(smx_surfel_buffers_cpu), sizeof(smx_recon_stats), sizeof(smx_driver_config), sizeof(smx_driver_step),\n'
        '  sizeof(smx_driver_host_frame), sizeof(smx_surfel_delta_cpu)); return 0; }\n')
    exe = tmp_path / "abi_probe"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    from surfelmeshing_amd._lib import BufferDesc, IntegrateParams, ReconStats, SurfelBuffersCPU, SurfelDeltaCPU
    from surfelmeshing_amd.pipeline import DriverConfig, DriverHostFrame, DriverStep
    assert sizes == [ctypes.sizeof(BufferDesc), ctypes.sizeof(IntegrateParams), ctypes.sizeof(SurfelBuffersCPU),
                     ctypes.sizeof(ReconStats), ctypes.sizeof(DriverConfig), ctypes.sizeof(DriverStep),
                     ctypes.sizeof(DriverHostFrame), ctypes.sizeof(SurfelDeltaCPU)]


def test_library_leaves_the_environment_alone_and_reports_what_it_wants():
    """Rounds 4-5 raised GPU_MAX_HW_QUEUES from a constructor of libsmx.so (process-global state changed behind the host's
    back; advisor r5).  Now loading the library changes nothing; smx_runtime_advice() says what the application should set,
    and the Python binding / bench.py (the applications here) set it themselves before the HIP runtime starts.  Checked in
    fresh processes at the C level (os.environ is a snapshot)."""
    import subprocess
    import sys
    from surfelmeshing_amd import _lib
    code = ("import ctypes, os, sys\n"
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
            "L = ctypes.CDLL(sys.argv[1])\n"
            "v = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
            "buf = ctypes.create_string_buffer(512)\n"
            "L.smx_runtime_advice.argtypes = [ctypes.c_char_p, ctypes.c_size_t]\n"
            "n = L.smx_runtime_advice(buf, 512)\n"
            "print(v.decode() if v else 'unset', n, buf.value.decode())\n")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    r = subprocess.run([sys.executable, "-c", code, _lib.SO_PATH], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.startswith("unset 1 GPU_MAX_HW_QUEUES is unset"), (r.stdout, r.stderr[-500:])
    r = subprocess.run([sys.executable, "-c", code, _lib.SO_PATH], capture_output=True, text=True, env=dict(env, GPU_MAX_HW_QUEUES="2"))
    assert r.returncode == 0 and r.stdout.startswith("2 1 GPU_MAX_HW_QUEUES is 2"), (r.stdout, r.stderr[-500:])
    r = subprocess.run([sys.executable, "-c", code, _lib.SO_PATH], capture_output=True, text=True, env=dict(env, GPU_MAX_HW_QUEUES="8"))
    assert r.returncode == 0 and r.stdout.strip() == "8 0", (r.stdout, r.stderr[-500:])
    # the Python binding is an application: importing it sets the default before torch / HIP start
    code = "import os; from surfelmeshing_amd import _lib; print(os.environ.get('GPU_MAX_HW_QUEUES'))"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=os.path.dirname(os.path.dirname(_lib.SO_PATH)))
    assert r.returncode == 0 and r.stdout.strip() == "8", (r.stdout, r.stderr[-500:])
