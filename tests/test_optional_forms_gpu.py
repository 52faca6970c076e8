"""The two optional weight-gradient forms of round 6 (conv_wgrad_split.hip: UZ_WG9=1 - four waves, nine taps per wave, two workgroups per CU;
UZ_WG_M16=1 - the 16x16x32 MFMA shape; UZ_WG_DB=1 - two LDS images; UZ_WG_HALF=1 - half workgroups, 64 x 32 tiles).  All are OFF in the product (measured slower, profiles/NOTES_r6.md 8) but stay in the library:
this keeps them correct.  The switches are read once per process, so each form runs in a child process (tools/wgrad_forms_check.py)
against an fp64 reference of autograd's weight gradient (reference: torchlayers.py:18, nn.Conv2d)."""
import os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env", [{"UZ_WG9": "1"}, {"UZ_WG_M16": "1"}, {"UZ_WG_DB": "1"}, {"UZ_WG_HALF": "1"}, {"UZ_WG9": "0", "UZ_WG_M16": "0"}],
                         ids=["nine_taps", "mfma16", "two_images", "half_workgroups", "product"])
def test_optional_weight_gradient_forms_vs_fp64(env):
    e = dict(os.environ); e.update(env)
    if e.get("UZ_CONV_MATH", "") in ("f32", "0", "bf16", "3"):
        pytest.skip("the split-fp16 weight gradient is not in play in this arithmetic mode")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wgrad_forms_check.py")], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "ALL OK" in r.stdout, r.stdout[-2000:]


@pytest.mark.parametrize("px", ["2048", "0"], ids=["lds_free", "product"])
def test_lds_free_small_plane_convolution_vs_fp64(px):
    """conv_mfma.hip conv_free_kernel (UZ_CONV_FREE_PX; off in the product): forward and data gradient of the 2 x 2 ... 8 x 8 planes,
    ragged channels, channel-slice views, accumulate - tools/conv_free_check.py, in a child process (the switch is read once)."""
    e = dict(os.environ); e["UZ_CONV_FREE_PX"] = px
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "conv_free_check.py")], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
