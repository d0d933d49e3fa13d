"""CPU-side guard for the round-1 "intermittent wrong rows" fix (DESIGN.md 3c'): the fault was a
packed-fp32 VALU instruction (`v_pk_mul_f32`, emitted by the SLP vectoriser) inside a matrix-core
loop, and the fix is a compiler flag (`-fno-slp-vectorize`, csrc/Makefile).  Nothing but this test
stops a compiler bump, a dropped flag or a hand-written float2 expression from putting one back:
it compiles the matrix-core translation units to gfx950 assembly with the library's own flags
(`make asm`) and fails if any kernel holds both a `v_mfma_*` and a `v_pk_{mul,add,fma}_f32`.
No GPU needed (hipcc cross-compiles)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "language_interpolation_amd", "csrc")
UNITS = ["pp_bf3_fwd", "pp_bf3_bwdx", "pp_bf3_bwdw", "pp_dense"]
PACKED = re.compile(r"^\s*(v_pk_(?:mul|add|fma)_f32)\b", re.M)


def kernels_of(asm_text):
    """(name, body) of every kernel in a gfx950 .s file: from its `name:` label to the
    `.Lfunc_endN:` label that closes the function (a kernel may hold several s_endpgm)."""
    names = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm_text, re.M)
    out = []
    for name in names:
        m = re.search(r"^%s:[^\n]*\n(.*?)^\.Lfunc_end\d+:" % re.escape(name), asm_text, re.S | re.M)
        assert m, f"no body found for kernel {name}"
        out.append((name, m.group(1)))
    return out


def test_no_packed_fp32_instruction_in_any_matrix_core_kernel(tmp_path_factory):
    asmdir = os.environ.get("PP_ASMDIR") or "/tmp/pphip_asm"
    subprocess.run(["make", "-C", CSRC, "asm", f"ASMDIR={asmdir}", "-j", "8"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    seen_mfma = 0
    offenders = []
    for unit in UNITS:
        text = open(os.path.join(asmdir, unit + ".s")).read()
        ks = kernels_of(text)
        assert ks, unit
        for name, body in ks:
            if "v_mfma_" not in body:
                continue
            seen_mfma += 1
            hits = PACKED.findall(body)
            if hits:
                offenders.append((unit, name, sorted(set(hits)), len(hits)))
    assert seen_mfma >= 20, seen_mfma  # the parser did look at the matrix-core kernels
    assert not offenders, offenders


def test_the_guard_recognises_a_packed_instruction():
    fake = """\t.amdhsa_kernel k1\n\t.end_amdhsa_kernel\nk1:\n\tv_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[4:7], a[0:15]\n\tv_pk_mul_f32 v[48:49], v[92:93], v[98:99] op_sel:[0,1]\n\ts_endpgm\n.Lfunc_end0:\n"""
    (name, body), = kernels_of(fake)
    assert name == "k1" and "v_mfma_" in body and PACKED.findall(body) == ["v_pk_mul_f32"]
