"""GPU: a short run of tools/soak_parity.py --against reference -- random 0.3-24 M-element cases (fp32 / fp64, every mode, ±inf /
NaN, a share with extreme parameters) whose expected values come from the REFERENCE's own CPU ops at run time:
oracle/_ref/libtorchlsq_ref_ops.so, built from the reference's four CPU translation units by oracle/build_ref.py in the build
container and shipped with the snapshot (nothing here reads /root/reference).  y, dx bit for bit; d_scale / d_shift inside
the bar; the oracle is held to the same values in the same run.  The committed 10-minute run: profiles/r05_soak_vs_reference.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libtorchlsq_ref_ops.so")


@pytest.mark.skipif(not os.path.isfile(REF_SO), reason="oracle/_ref/libtorchlsq_ref_ops.so was not built (no /root/reference at build time)")
def test_random_policy_sized_cases_against_the_reference_cpu_ops():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_parity.py"), "--minutes", "0.5", "--seed", "7",
                        "--against", "reference", "--extreme", "0.15"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("cases ")][0].split()
    assert int(line[1]) >= 20 and int(line[-1]) == 0, r.stdout[-2000:]
    assert "REFERENCE's CPU ops" in r.stdout
