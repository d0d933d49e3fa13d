"""Generator of the inline-assembly body of pp_bwd_weight_regacc_kernel (pp_regacc.hip).

python tools/gen_regacc_asm.py > language_interpolation_amd/csrc/pp_regacc_body.inc

The kernel keeps one accumulator per weight-table row in a VGPR (lane <-> out) and adds each
batch row into the register picked by the row's segment, with the VGPR index mode of gfx9
(s_set_gpr_idx_on / s_set_gpr_idx_idx: M0[7:0] is added to the register number of the operands
selected by the mode bits).  The compiler cannot express that (a dynamically indexed array of
this size goes to scratch; smaller ones cost 7 instructions per update), so the loop is written
here.  Register map (everything from s24 / v8 up is listed as clobbered; s32/s33 are left alone,
the ABI reserves them):

  s[24:27] buffer resource of gy (base advanced / num_records reduced per block of 16 rows, so
           rows past the end of the batch read 0)
  s[28:29] destination row pointer       s[30:31] record pointer (per 64 rows: P planes of 64)
  s34 bytes per 16 rows of gy            s35 blocks left, then the row counter of the write-out
  s[36:36+16P) the scalars of the 16 rows of the block being accumulated (k0, then basis values)
  v[8:8+P) records of the current 64 rows (lane <-> row), v[12:12+P) of the next 64
  v[16:31] byte offset of row r of a block + this lane's out
  v[32:ACC0) ring of D buffers of 16 gy values     v[ACC0:255] accumulators, ACC0 = 256 - KMAX
  (two variants: KMAX = 160 with 64 staging registers, D = 4; KMAX = 192 with 32, D = 2)

One phase = one block of 16 rows: read the block's scalars out of the record registers
(v_readlane), start the gy loads of the block D-1 ahead, wait for this block's gy, accumulate
(index mode on only here).  Four phases = 64 rows = one set of record registers; the records
of the next 64 rows are loaded at phase 0 into the other set (the loop body holds 8 phases).  A first version
read the scalars with s_load one block ahead and ran at the scalar-cache miss latency (17 ns a
row for n = 1, 43 ns for n = 2: 700 cycles per block); records travelling through VGPRs are
prefetched 64 rows ahead for P registers.  P = 1 for n = 1 (the basis value is 1; rows past the
batch add the 0 the exhausted buffer resource returns), 1 + n otherwise.

Measured on the cfg3 hidden layer (n = 1, 200 -> 200, 150 segments, 16384 rows; 192 us): the
loop is at two limits at once.  With the gy loads removed it takes 170 us (v_readlane costs 7-8
cycles, the accumulate 5, the per-block skeleton with its mode switches 130 cycles); with only
the loads left 180 us (every wave-row is a 256-byte request that straddles three 128-byte lines
of the 800-byte rows: the L1 path of a CU moves 64 bytes a cycle).  Leaving the index mode on
for the whole loop makes every VALU instruction slower (236 us).
"""
import sys

REC_CUR = 8
REC_NXT = 12
VOFF0 = 16   # v[16:31]: per-row byte offsets
STG0 = 32    # staging starts here, accumulators follow
RB = 16
SC0 = 36     # scalars of the current block


def body(N, KMAX):
    P = 1 if N == 1 else 1 + N
    ACC0 = 256 - KMAX
    D = (ACC0 - STG0) // RB          # ring of D gy buffers, loads run D-1 blocks ahead
    assert D in (2, 4) and SC0 + P * RB <= 100
    L = []
    a = L.append

    def rec_loads(base):
        for j in range(P):
            a(f"global_load_dword v{base + j}, %[lane4], s[30:31] offset:{j * 256}")
        a(f"s_add_u32 s30, s30, {P * 256}")
        a("s_addc_u32 s31, s31, 0")

    def vmem(slot):
        for r in range(RB):
            a(f"buffer_load_dword v{STG0 + slot * RB + r}, v{VOFF0 + r}, s[24:27], 0 offen")
        a("s_add_u32 s24, s24, s34")
        a("s_addc_u32 s25, s25, 0")
        a("s_sub_i32 s26, s26, s34")
        a("s_max_i32 s26, s26, 0")

    def readlanes(ph, base):
        for j in range(P):
            for r in range(RB):
                a(f"v_readlane_b32 s{SC0 + j * RB + r}, v{base + j}, {ph * RB + r}")

    def process(slot):
        # index mode on only around the accumulates: with it on, every VALU instruction (the
        # v_readlane's too) issues slower (measured: 236 vs 192 us on the cfg3 hidden layer)
        mode = "SRC1,DST" if N == 1 else "SRC2,DST"
        for r in range(RB):
            if r == 0:
                a(f"s_set_gpr_idx_on s{SC0}, gpr_idx({mode})")
            else:
                a(f"s_set_gpr_idx_idx s{SC0 + r}")
            g = STG0 + slot * RB + r
            if N == 1:
                a(f"v_add_f32 v{ACC0}, v{g}, v{ACC0}")
            else:
                for j in range(N):
                    a(f"v_fma_f32 v{ACC0 + j}, s{SC0 + (1 + j) * RB + r}, v{g}, v{ACC0 + j}")
        a("s_set_gpr_idx_off")

    # ---- set-up
    a("s_mov_b32 s24, %[r0]")
    a("s_mov_b32 s25, %[r1]")
    a("s_mov_b32 s26, %[r2]")
    a("s_mov_b32 s27, %[r3]")
    a("s_mov_b64 s[28:29], %[dst]")
    a("s_mov_b64 s[30:31], %[rec]")
    a("s_mov_b32 s35, %[nblk]")
    a(f"s_mul_i32 s34, %[rowb], {RB}")
    a(f"v_mov_b32 v{VOFF0}, %[laneb]")
    for r in range(1, RB):
        a(f"v_add_u32 v{VOFF0 + r}, %[rowb], v{VOFF0 + r - 1}")
    for k in range(KMAX):
        a(f"v_mov_b32 v{ACC0 + k}, 0")
    # ---- prologue: records of the first 64 rows, gy of blocks 0..D-2
    rec_loads(REC_CUR)
    for t in range(D - 1):
        vmem(t)
    a("1:")
    for ph8 in range(8):
        ph = ph8 & 3
        cur = REC_CUR if ph8 < 4 else REC_NXT
        nxt = REC_NXT if ph8 < 4 else REC_CUR
        if ph == 0:
            a(f"s_waitcnt vmcnt({(D - 1) * RB})")
        readlanes(ph, cur)
        vmem((ph + D - 1) % D)
        if ph == 0:
            rec_loads(nxt)
        a(f"s_waitcnt vmcnt({(D - 1) * RB + (P if ph <= D - 1 else 0)})")
        process(ph % D)
        a("s_sub_u32 s35, s35, 1")
        a("s_cmp_eq_u32 s35, 0")
        if ph8 < 7:
            a("s_cbranch_scc1 2f")
        else:
            a("s_cbranch_scc0 1b")
    a("2:")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    # ---- write-out: row k of the table = accumulator k, lanes with an out only
    a("v_cmp_ne_u32 vcc, 0, %[vok]")
    a("s_and_b64 exec, exec, vcc")
    a("s_mov_b32 s35, 0")
    a("3:")
    a("s_set_gpr_idx_on s35, gpr_idx(SRC1)")
    a(f"v_mul_f32 v{STG0}, %[scale], v{ACC0}")
    a("s_set_gpr_idx_off")
    a(f"global_store_dword v{VOFF0}, v{STG0}, s[28:29]")
    a("s_add_u32 s28, s28, %[rowb]")
    a("s_addc_u32 s29, s29, 0")
    a("s_add_u32 s35, s35, 1")
    a("s_cmp_lt_u32 s35, %[K]")
    a("s_cbranch_scc1 3b")
    a("s_mov_b64 exec, -1")
    return L


def main():
    out = sys.stdout
    out.write("// Generated by tools/gen_regacc_asm.py -- do not edit; see that file for the register map.\n")
    for N in (1, 2, 3):
        for KMAX in (160, 192):
            out.write(f"#define PP_REGACC_BODY_N{N}_K{KMAX} \\\n")
            for line in body(N, KMAX):
                out.write(f'  "{line}\\n" \\\n')
            out.write('  ""\n')
    regs = [f'"s{i}"' for i in range(24, 100) if i not in (32, 33)] + [f'"v{i}"' for i in range(8, 256)]
    out.write("#define PP_REGACC_CLOBBERS \\\n")
    for i in range(0, len(regs), 12):
        out.write("  " + ", ".join(regs[i:i + 12]) + ", \\\n")
    out.write('  "vcc", "scc", "memory"\n')
    out.write("#define PP_REGACC_KMAX 192\n")
    out.write("#define PP_REGACC_KDEEP 160\n")


if __name__ == "__main__":
    main()
