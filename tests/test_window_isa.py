"""The window kernel's LDS reads are inline-asm ``ds_read_*`` the compiler's waitcnt pass cannot see; the one
``s_waitcnt lgkmcnt(0)`` behind them is tied to their destinations through "+v" constraints only.  A compiler
upgrade that put a copy or a spill of such a destination between a read and the wait would read the register before
the LDS has answered - silently wrong numbers.  This test compiles ``csrc/semilag.hip`` to gfx950 assembly (no GPU
needed) and fails if, anywhere in a ``semilag_window`` kernel, an instruction touches a register a read still in flight
writes (reads complete in order: ``s_waitcnt lgkmcnt(N)`` leaves the N youngest pending), if a basic block ends with reads pending,
or if the default kernel uses scratch memory (a VGPR spill inside the 128-register budget of four waves per SIMD)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pysteps_amd", "csrc", "semilag.hip")


def _regs(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(r) for r in re.findall(r"\bv(\d+)\b", text))
    return out


@pytest.fixture(scope="module")
def window_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "semilag.s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, SRC],
                   check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def _kernels(asm):
    """name -> (instruction lines, metadata block) of every semilag_window instantiation"""
    found = {}
    for m in re.finditer(r"^(_ZN3psh\S*semilag_window\S*):.*?\n(.*?)\.end_amdhsa_kernel", asm, re.S | re.M):
        found[m.group(1)] = m.group(2)
    return found


def test_no_instruction_touches_a_pending_lds_destination(window_asm):
    kernels = _kernels(window_asm)
    assert len(kernels) >= 2, list(kernels)
    groups = 0
    for name, body in kernels.items():
        pending, in_asm = [], False  # destination sets of the reads in flight, oldest first (the LDS answers in order)
        for line in body.splitlines():
            code = line.split(";")[0].strip() if not line.lstrip().startswith(";;#") else line.strip()
            if code.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if code.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not code:
                continue
            if re.match(r"^\.?L?BB\d+_\d+:", code) or code.endswith(":"):
                assert not pending, "%s: basic block boundary with LDS reads pending: %s" % (name, line)
                continue
            if code.startswith("."):
                continue
            if in_asm and code.startswith("ds_read"):
                dest = code.split(",")[0]
                pending.append(_regs(dest))
                continue
            waited = re.search(r"s_waitcnt.*lgkmcnt\((\d+)\)", code)
            if waited:
                # at most N operations outstanding afterwards: everything but the N youngest reads has landed (the
                # counter saturates at 15 - more reads than that cannot be in flight, the sequencer stalls the issue)
                keep = int(waited.group(1))
                if pending and keep == 0:
                    groups += 1
                pending = pending[len(pending) - keep:] if keep < len(pending) else pending
                if keep == 0:
                    pending = []
                continue
            if pending:
                in_flight = set().union(*pending)
                assert not code.startswith(("s_cbranch", "s_branch", "s_endpgm")), (name, line)
                assert not code.startswith(("scratch_", "buffer_store")) or not (_regs(code) & in_flight), (name, line)
                hit = _regs(code) & in_flight
                assert not hit, "%s: `%s` touches v%s while its ds_read is in flight" % (name, code, sorted(hit))
    assert groups >= 6  # three sampling passes per kernel at least


def test_default_window_kernel_has_no_scratch(window_asm):
    # the two instantiations a default call takes (boundary mode "constant" / the others), one tile per workgroup
    blocks = re.findall(r"\.name:\s+(\S*semilag_window\S*ILi8ELi96ELi64ELi4ELi4EEELb[01]ELb0E\S*)\n(.*?)\.wavefront_size",
                        window_asm, re.S)
    assert len(blocks) == 2, [b[0] for b in blocks]
    for name, meta in blocks:
        assert re.search(r"\.private_segment_fixed_size:\s+0\b", meta), (name, "scratch in use")
        assert re.search(r"\.vgpr_spill_count:\s+0\b", meta), (name, "VGPR spills")
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)) <= 128, name
