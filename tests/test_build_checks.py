"""Build-time checks of the persistent decode launches (csrc/engine.hip) that need no GPU: hipcc cross-compiles gfx950 here.

The streaming waves keep a ring of weight batches in FIXED physical registers across inline-assembly statements
(DESIGN.md 4.6): a load issued by one statement lands while later statements run, so the compiler must never park a
value of its own in those registers between two statements.  The statements clobber them; scripts/check_engine_regs.py
reads hipcc's ISA and verifies nothing outside the statements touches them.  The statements' text is generated
(scripts/gen_engine_ring.py): the committed engine_ring.inc must be what the generator writes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "llama2.zig_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def test_engine_ring_inc_is_the_generators_output():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_engine_ring.py")], stdout=subprocess.PIPE, check=True).stdout
    assert out.decode() == open(os.path.join(CSRC, "engine_ring.inc")).read(), \
        "engine_ring.inc is stale: python scripts/gen_engine_ring.py > llama2.zig_amd/csrc/engine_ring.inc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_hipcc_stays_off_the_engines_register_ring(tmp_path):
    isa = str(tmp_path / "engine.s")
    # the Makefile's flags, device code only, as assembly
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-function",
                    "-S", "--cuda-device-only", "-I", CSRC, "-o", isa, os.path.join(CSRC, "engine.hip")],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_engine_regs.py"), isa], stdout=subprocess.PIPE)
    assert r.returncode == 0, r.stdout.decode()
    # and no spills: a spilled ring register would be reloaded from a stale copy
    import re
    text = open(isa).read()
    body = text[re.search(r"^_ZN\S*engine_kernel\S*:", text, re.M).start():]
    m = re.search(r"; ScratchSize: (\d+)", body)
    assert m and int(m.group(1)) == 0, f"the persistent kernel uses {m.group(1) if m else '?'} bytes of scratch"
