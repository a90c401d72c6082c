"""The build-time guard against gfx950's 64-bit-shift erratum (profiles/r03_shift64_erratum.md): no kernel of the shipped library
may hold the amount of a v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 in the last VGPR of its allocation.  CPU: the scanner reads
the ISA out of the built library."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCANNER = os.path.join(ROOT, "tools", "scan_shift64_top_vgpr.py")


def _scanner():
    spec = importlib.util.spec_from_file_location("scan_shift64_top_vgpr", SCANNER)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_KERNEL = """
	.text
_Z6kernelPm:
	s_load_dwordx2 s[0:1], s[0:1], 0x0
	%s
	s_endpgm
	.amdhsa_kernel _Z6kernelPm
		.amdhsa_next_free_vgpr %d
		.amdhsa_accum_offset %d
	.end_amdhsa_kernel
"""


def _hits(tmp_path, instruction, vgprs):
    path = tmp_path / "k.s"
    path.write_text(_KERNEL % (instruction, vgprs, (vgprs + 3) // 4 * 4))
    hits, kernels = _scanner().scan_asm(str(path))
    assert kernels == 1
    return len(hits)


def test_scanner_recognises_the_shape(tmp_path):
    # the library kernel's own instruction: amount in v31, 32 registers in use
    assert _hits(tmp_path, "v_lshrrev_b64 v[8:9], v31, v[8:9]", 32) == 1
    assert _hits(tmp_path, "v_lshlrev_b64 v[8:9], v31, v[8:9]", 32) == 1
    assert _hits(tmp_path, "v_ashrrev_i64 v[8:9], v23, v[8:9]", 24) == 1
    assert _hits(tmp_path, "v_lshrrev_b64 v[8:9], v39, v[8:9]", 40) == 1
    # not the last register of the allocation, or the next one is in use, or the amount is no VGPR
    assert _hits(tmp_path, "v_lshrrev_b64 v[8:9], v30, v[8:9]", 32) == 0
    assert _hits(tmp_path, "v_lshrrev_b64 v[8:9], v31, v[8:9]", 33) == 0          # (what KATOME_SHIFT64_GUARD(32) does)
    assert _hits(tmp_path, "v_lshrrev_b64 v[8:9], v31, v[8:9]", 40) == 0
    assert _hits(tmp_path, "v_lshrrev_b64 v[8:9], s13, v[8:9]", 32) == 0
    assert _hits(tmp_path, "v_lshrrev_b64 v[8:9], 1, v[8:9]", 32) == 0
    assert _hits(tmp_path, "v_lshrrev_b32_e32 v31, 1, v31", 32) == 0              # a 32-bit shift


def test_scanner_counts_architectural_registers_only(tmp_path):
    """gfx950's register file is unified: .amdhsa_next_free_vgpr counts the AGPRs behind .amdhsa_accum_offset too, and the erratum
    is about the last ARCH register -- a kernel that spills into AGPRs, or whose allocation has a hole behind v(8n+7), has the shape"""
    kernel = """
	.text
_Z6kernelPm:
	%s
	s_endpgm
	.amdhsa_kernel _Z6kernelPm
		.amdhsa_next_free_vgpr %d
		.amdhsa_accum_offset %d
	.end_amdhsa_kernel
"""
    def hits(body, total, accum):
        path = tmp_path / "a.s"
        path.write_text(kernel % (body, total, accum))
        found, kernels = _scanner().scan_asm(str(path))
        assert kernels == 1
        return len(found)
    shift = "v_lshrrev_b64 v[8:9], v31, v[8:9]"
    assert hits(shift + "\n\tv_accvgpr_write_b32 a0, v3", 40, 32) == 1            # 32 arch registers + 8 AGPRs: v31 is the last arch one
    assert hits(shift + "\n\tv_mov_b32_e32 v32, 0", 40, 40) == 0                  # v32 is named: the allocation goes on
    # a hole at v32 (nothing names it) inside an allocation that goes on: the hardware's condition is the allocation's end, so no hit --
    # unless LLVM's stricter rule is asked for (KATOME_SCAN_STRICT=1: "the next register is named by no instruction")
    assert hits(shift + "\n\tv_mov_b32_e32 v33, 0", 40, 40) == 0
    mod = _scanner()
    assert mod.is_hit(31, 40, {0, 31, 33}) and not mod.is_hit(31, 40, {0, 31, 32}) and not mod.is_hit(31, 40, None)


def test_scanner_recognises_a_dropped_masked_assignment(tmp_path):
    """profiles/r04_wrong_code.md: hipcc 7.2 dropped `ne = 2` under `uniform || divergent` in lds_count_wide_kernel's read-out and left
    an empty masked region behind -- `s_and_saveexec_b64 sX, cond` followed at once by `s_or_b64 exec, exec, sX`"""
    kernel = """
	.text
_Z6kernelPm:
	v_cmp_ne_u64_e32 vcc, v[8:9], v[18:19]
	s_and_saveexec_b64 s[12:13], vcc
%s	s_or_b64 exec, exec, s[12:13]
	s_endpgm
	.amdhsa_kernel _Z6kernelPm
		.amdhsa_next_free_vgpr 24
		.amdhsa_accum_offset 24
	.end_amdhsa_kernel
"""
    def hits(body):
        path = tmp_path / "m.s"
        path.write_text(kernel % body)
        return len(_scanner().scan_asm(str(path))[0])
    assert hits("") == 1
    assert hits("; %bb.1:\n") == 1                                   # (comments and labels of the listing in between do not hide it)
    assert hits("\tv_mov_b32_e32 v4, 2\n") == 0                      # the assignment is there: fine


def test_shipped_library_is_clear_of_the_shape():
    lib = os.path.join(ROOT, "katome_amd", "lib", "libkatome_gpu.so")
    assert os.path.exists(lib), "build the library first (__graft_entry__.build())"
    out = subprocess.run([sys.executable, SCANNER, lib], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    n_kernels = int(last.split()[0])
    assert n_kernels > 200 and " 0 64-bit shifts" in last and "empty masked regions" in last, last       # every code object of the library was read
