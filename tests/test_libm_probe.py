"""csrc/hrt_libm.h (the device's sinf/cosf/expf/acosf) compiled for the host and compared with
the host libm: sampled here (every 1009th float of each domain, ~2 s); `oracle/libm_probe
--full` is the exhaustive run recorded in DESIGN.md."""
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_math_header_equals_host_libm():
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "libm_probe"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(REPO, "oracle", "libm_probe")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "OK" in out.stdout
