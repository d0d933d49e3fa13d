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

NO_LOADS = "--no-loads" in sys.argv
LDS_NOWAIT = "--lds-nowait" in sys.argv        # developer ablations of body_lds (wrong values, same work)
LDS_NOBARRIER = "--lds-nobarrier" in sys.argv
REC_CUR = 8
REC_NXT = 12
VOFF0 = 16   # v[16:31]: per-row byte offsets
STG0 = 32    # staging starts here, accumulators follow
RB = 16
SC0 = 36     # scalars of the current block


def write_out(a, KMAX, ACC0, voff):
    """Row k of the table = accumulator k, lanes with an out only.  Unrolled over the registers
    (round 3): the first version walked them with the index mode (on / v_mul / off / store per
    row), ~13 us of fixed cost in a launch at the reference's batch sizes."""
    a("v_cmp_ne_u32 vcc, 0, %[vok]")
    a("s_and_b64 exec, exec, vcc")
    a("s_cbranch_execz 4f")
    for k in range(KMAX):
        a(f"s_cmp_le_u32 %[K], {k}")
        a("s_cbranch_scc1 4f")
        a(f"v_mul_f32 v{ACC0 + k}, %[scale], v{ACC0 + k}")
        a(f"global_store_dword v{voff}, v{ACC0 + k}, s[28:29]")
        a("s_add_u32 s28, s28, %[rowb]")
        a("s_addc_u32 s29, s29, 0")
    a("4:")
    a("s_mov_b64 exec, -1")


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
            if not NO_LOADS:  # (--no-loads: developer variant, what the loop costs without its gy stream)
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
    write_out(a, KMAX, ACC0, VOFF0)
    return L


def body_lds(N, KMAX, PK=False):
    """Variant with gy shared through LDS (round 3).  The four waves of a workgroup (four inputs,
    same rows, same 64 outs) each loaded every gy row themselves: 3.4 GB through the vector-memory
    path per launch on the cfg3 hidden layer, which is what its 180-190 us were (without the loads
    the loop takes 134 us at one wave per SIMD and 90 us at two; profiles/r03_e_regacc_lds.txt).
    Here a block of 64 rows x 256 bytes is brought into one of two LDS buffers ONCE per workgroup by
    LDS-DMA (buffer_load_dwordx4 ... lds, four instructions per wave, the buffer resource still
    returns 0 past the end of the batch) while the waves accumulate the previous block out of the
    other buffer with ds_read2st64_b32 (address = lane*4, two rows an instruction, rows and buffer in
    the immediate offsets; 119 -> 106 us against one ds_read_b32 a row).  One
    barrier per 64 rows.  Register map as body() except:
      v16 store offset (lane's out)   v17 LDS read address (lds base + lane*4)
      v[20:23] DMA source offsets of this wave's four pieces (rows 4(w+4t)..+3, t = 0..3)
      v[32:39], v[40:47] two sets of 8 gy values (first / second half of a block of 16 rows)
      s34 bytes per 64 rows of gy     s22 scratch     s23 LDS address of this wave's first DMA piece
    PK (n = 2, discontinuous: k0 even, so the two accumulators of a row are an aligned register
    pair): ONE indexed v_pk_fma_f32 a row instead of two v_fma_f32 -- the indexed VALU operation
    is what the loop is bound by (8 cycles each, measured on the n = 1 body).  The basis values
    of row r sit in the aligned pair s[52+2r:53+2r]; gy is the low or the high half of an aligned
    staging pair (op_sel).
    """
    P = 1 if N == 1 else 1 + N
    ACC0 = 256 - KMAX
    assert ACC0 >= 48 and SC0 + P * RB <= 100 and (not PK or N == 2)
    SA, SB_ = 32, 40
    L = []
    a = L.append

    def rec_loads(base):
        for j in range(P):
            a(f"global_load_dword v{base + j}, %[lane4], s[30:31] offset:{j * 256}")
        a(f"s_add_u32 s30, s30, {P * 256}")
        a("s_addc_u32 s31, s31, 0")

    def dma(buf):
        # this wave's four 1-KiB pieces of the NEXT block of 64 rows, then advance the resource
        for t in range(4):
            a(f"s_add_u32 m0, s23, {buf * 16384 + t * 4096}")
            a("s_nop 0")
            a(f"buffer_load_dwordx4 v{20 + t}, s[24:27], 0 offen lds")
        a("s_add_u32 s24, s24, s34")
        a("s_addc_u32 s25, s25, 0")
        a("s_sub_i32 s26, s26, s34")
        a("s_max_i32 s26, s26, 0")

    def lds_reads(buf, row0, stg):
        # rows r and r+1 of the block are 256 bytes apart = one "st64" unit: one instruction a pair
        for r in range(0, 8, 2):
            a(f"ds_read2st64_b32 v[{stg + r}:{stg + r + 1}], v17 offset0:{buf * 64 + row0 + r} offset1:{buf * 64 + row0 + r + 1}")

    def readlanes(ph, base):
        for j in range(P):
            for r in range(RB):
                dst = SC0 + j * RB + r if not (PK and j) else SC0 + RB + 2 * r + j - 1
                a(f"v_readlane_b32 s{dst}, v{base + j}, {ph * RB + r}")

    def accumulate(r, g):
        if PK:
            m = SC0 + RB + 2 * r
            if g % 2 == 0:
                a(f"v_pk_fma_f32 v[{ACC0}:{ACC0 + 1}], s[{m}:{m + 1}], v[{g}:{g + 1}], v[{ACC0}:{ACC0 + 1}] op_sel_hi:[1,0,1]")
            else:
                a(f"v_pk_fma_f32 v[{ACC0}:{ACC0 + 1}], s[{m}:{m + 1}], v[{g - 1}:{g}], v[{ACC0}:{ACC0 + 1}] op_sel:[0,1,0] op_sel_hi:[1,1,1]")
        elif N == 1:
            a(f"v_add_f32 v{ACC0}, v{g}, v{ACC0}")
        else:
            for j in range(N):
                a(f"v_fma_f32 v{ACC0 + j}, s{SC0 + (1 + j) * RB + r}, v{g}, v{ACC0 + j}")

    def process():
        mode = "SRC1,DST" if N == 1 else "SRC2,DST"
        a(f"s_set_gpr_idx_on s{SC0}, gpr_idx({mode})")
        a("s_waitcnt lgkmcnt(4)")
        for r in range(8):
            if r:
                a(f"s_set_gpr_idx_idx s{SC0 + r}")
            accumulate(r, SA + r)
        a("s_waitcnt lgkmcnt(0)")
        for r in range(8, 16):
            a(f"s_set_gpr_idx_idx s{SC0 + r}")
            accumulate(r, SB_ + r - 8)
        a("s_set_gpr_idx_off")

    # ---- set-up
    a("s_mov_b32 s24, %[r0]")
    a("s_mov_b32 s25, %[r1]")
    a("s_mov_b32 s26, %[r2]")
    a("s_mov_b32 s27, %[r3]")
    a("s_mov_b64 s[28:29], %[dst]")
    a("s_mov_b64 s[30:31], %[rec]")
    a("s_mov_b32 s35, %[nblk]")
    a("s_mov_b32 s23, %[ldsw]")
    a("s_lshl_b32 s34, %[rowb], 6")
    a("v_mov_b32 v16, %[laneb]")
    a("v_mov_b32 v17, %[ldsr]")
    a("v_mov_b32 v20, %[dma0]")
    a("s_lshl_b32 s22, %[rowb], 4")
    for t in range(1, 4):
        a(f"v_add_u32 v{20 + t}, s22, v{20 + t - 1}")
    for k in range(KMAX):
        a(f"v_mov_b32 v{ACC0 + k}, 0")
    # ---- prologue: records and gy of the first 64 rows
    rec_loads(REC_CUR)
    dma(0)
    if LDS_NOWAIT:
        a("s_waitcnt vmcnt(0)")
    a("1:")
    for ph8 in range(8):
        ph = ph8 & 3
        buf = ph8 >> 2
        cur = REC_CUR if ph8 < 4 else REC_NXT
        nxt = REC_NXT if ph8 < 4 else REC_CUR
        if ph == 0:
            if LDS_NOWAIT:
                a("s_waitcnt vmcnt(4)")
            else:
                a("s_waitcnt vmcnt(0)")   # this block's records and DMA pieces (issued 64 rows ago)
            if not LDS_NOBARRIER:
                a("s_barrier")            # ... everyone's; and everyone is done with the other buffer
            rec_loads(nxt)
            dma(buf ^ 1)
            lds_reads(buf, 0, SA)
        readlanes(ph, cur)
        lds_reads(buf, ph * RB + 8, SB_)
        process()
        if ph < 3:
            lds_reads(buf, (ph + 1) * RB, SA)
        a("s_sub_u32 s35, s35, 1")
        a("s_cmp_eq_u32 s35, 0")
        if ph8 < 7:
            a("s_cbranch_scc1 2f")
        else:
            a("s_cbranch_scc0 1b")
    a("2:")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    write_out(a, KMAX, ACC0, 16)
    return L


X_NODS = "--x-nods" in sys.argv      # developer ablations of body_lds1 (wrong values, in-range indices)
X_NODMA = "--x-nodma" in sys.argv
X_NOLSHR = "--x-nolshr" in sys.argv
X_NOIDX = "--x-noidx" in sys.argv
READ1 = "--read1" in sys.argv   # developer variant: one ds_read_b32 a row instead of ds_read2st64_b32 a pair
TMPIDX = "--tmpidx" in sys.argv  # developer variant: index bytes shifted into rotating temporaries three rows ahead


def body_lds1(KMAX):
    """n = 1 with gy through LDS (body_lds) and PACKED records: the only per-row scalar is the
    table row k0 < 256, so the records are one byte a row (rec8[i][row]) and one v_readlane brings
    four rows' indices into an SGPR; s_set_gpr_idx_idx takes bits 7:0, an s_lshr_b64 (SALU, off
    the vector pipe) moves the next byte of two registers down (106 -> 97 us against one
    s_lshr_b32 a row; the indices by s_load_dwordx16 instead of v_readlane: no difference).  v_readlane was 7-8 of the ~17 cycles a row of
    body_lds; here it is 16 per 64 rows, all issued at the top of the block, so the index mode
    stays on for the 64 accumulates of a block (the mode switches were the other large item).
    One record register holds 256 rows (v8; the next 256 are loaded into v12 a group ahead), the
    loop body is one group = four blocks of 64 rows.  s35 counts blocks of 64 rows.
    """
    ACC0 = 256 - KMAX
    assert ACC0 >= 48
    SA, SB_ = 32, 40
    IDX0 = 36  # s[36:51]: the 64 indices of a block, four to a register
    L = []
    a = L.append

    def dma(buf):
        for t in range(4):
            if X_NODMA:
                break
            a(f"s_add_u32 m0, s23, {buf * 16384 + t * 4096}")
            a("s_nop 0")
            a(f"buffer_load_dwordx4 v{20 + t}, s[24:27], 0 offen lds")
        a("s_add_u32 s24, s24, s34")
        a("s_addc_u32 s25, s25, 0")
        a("s_sub_i32 s26, s26, s34")
        a("s_max_i32 s26, s26, 0")

    def rec_load():
        a(f"global_load_dword v{REC_NXT}, %[lane4], s[30:31]")
        a("s_add_u32 s30, s30, 256")
        a("s_addc_u32 s31, s31, 0")

    def lds_reads(buf, row0, stg):
        for r in range(8):
            if X_NODS:
                continue
            if READ1:
                a(f"ds_read_b32 v{stg + r}, v17 offset:{buf * 16384 + (row0 + r) * 256}")
            elif r % 2 == 0:  # rows r and r+1 of the block: 256 bytes apart = one "st64" unit
                a(f"ds_read2st64_b32 v[{stg + r}:{stg + r + 1}], v17 offset0:{buf * 64 + row0 + r} offset1:{buf * 64 + row0 + r + 1}")

    a("s_mov_b32 s24, %[r0]")
    a("s_mov_b32 s25, %[r1]")
    a("s_mov_b32 s26, %[r2]")
    a("s_mov_b32 s27, %[r3]")
    a("s_mov_b64 s[28:29], %[dst]")
    a("s_mov_b64 s[30:31], %[rec]")
    a("s_mov_b32 s35, %[nblk]")
    a("s_mov_b32 s23, %[ldsw]")
    a("s_lshl_b32 s34, %[rowb], 6")
    a("v_mov_b32 v16, %[laneb]")
    a("v_mov_b32 v17, %[ldsr]")
    a("v_mov_b32 v20, %[dma0]")
    a("s_lshl_b32 s22, %[rowb], 4")
    for t in range(1, 4):
        a(f"v_add_u32 v{20 + t}, s22, v{20 + t - 1}")
    for k in range(KMAX):
        a(f"v_mov_b32 v{ACC0 + k}, 0")
    rec_load()
    dma(0)
    a("1:")
    for q in range(4):
        buf = q & 1
        a("s_waitcnt vmcnt(0)")   # this block's DMA pieces (and, q = 0, this group's records)
        if not X_NODMA:
            a("s_barrier")
        if q == 0:
            a(f"v_mov_b32 v{REC_CUR}, v{REC_NXT}")
            rec_load()
        dma(buf ^ 1)
        lds_reads(buf, 0, SA)
        for t in range(16):
            a(f"v_readlane_b32 s{IDX0 + t}, v{REC_CUR}, {16 * q + t}")
        a(f"s_set_gpr_idx_on s{IDX0}, gpr_idx(SRC1,DST)")
        for h in range(8):
            cur, oth = (SA, SB_) if h % 2 == 0 else (SB_, SA)
            if h < 7:
                lds_reads(buf, (h + 1) * 8, oth)
                a(f"s_waitcnt lgkmcnt({8 if READ1 else 4})")
            else:
                a("s_waitcnt lgkmcnt(0)")
            # the 8 rows of this half: bytes of the register pair s[lo:lo+1]; one 64-bit shift
            # moves the next byte of BOTH registers down, so the rows go 0,4,1,5,2,6,3,7
            lo = IDX0 + 2 * h
            if TMPIDX:
                # rows in natural order; byte r % 4 of register lo + r // 4 goes through an
                # independent 32-bit shift into s53..s55, issued three rows before its use
                def shift(r):
                    if r < 8 and r % 4:
                        a(f"s_lshr_b32 s{52 + r % 4}, s{lo + r // 4}, {8 * (r % 4)}")
                shift(1)
                shift(2)
                for r in range(8):
                    reg = lo + r // 4 if r % 4 == 0 else 52 + r % 4
                    if h or r:
                        a(f"s_set_gpr_idx_idx s{reg}")
                    a(f"v_add_f32 v{ACC0}, v{cur + r}, v{ACC0}")
                    shift(r + 3)
                continue
            for j in range(4):
                for w in range(2):
                    r = 4 * w + j
                    if h or j or w:
                        a(f"s_set_gpr_idx_idx s{lo + w}")
                    a(f"v_add_f32 v{ACC0}, v{cur + r}, v{ACC0}")
                if j < 3 and not X_NOLSHR:
                    a(f"s_lshr_b64 s[{lo}:{lo + 1}], s[{lo}:{lo + 1}], 8")
        a("s_set_gpr_idx_off")
        a("s_sub_u32 s35, s35, 1")
        a("s_cmp_eq_u32 s35, 0")
        if q < 3:
            a("s_cbranch_scc1 2f")
        else:
            a("s_cbranch_scc0 1b")
    a("2:")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    write_out(a, KMAX, ACC0, 16)
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
    for N in (1, 2, 3):
        for KMAX in (160, 192):
            out.write(f"#define PP_REGACC_LDS_BODY_N{N}_K{KMAX} \\\n")
            for line in body_lds(N, KMAX):
                out.write(f'  "{line}\\n" \\\n')
            out.write('  ""\n')
    for KMAX in (160, 192):
        out.write(f"#define PP_REGACC_LDS_BODY_N2PK_K{KMAX} \\\n")
        for line in body_lds(2, KMAX, PK=True):
            out.write(f'  "{line}\\n" \\\n')
        out.write('  ""\n')
    for KMAX in (160, 192):
        out.write(f"#define PP_REGACC_LDS1_BODY_K{KMAX} \\\n")
        for line in body_lds1(KMAX):
            out.write(f'  "{line}\\n" \\\n')
        out.write('  ""\n')
    regs = [f'"s{i}"' for i in range(22, 100) if i not in (32, 33)] + [f'"v{i}"' for i in range(8, 256)]
    out.write("#define PP_REGACC_CLOBBERS \\\n")
    for i in range(0, len(regs), 12):
        out.write("  " + ", ".join(regs[i:i + 12]) + ", \\\n")
    out.write('  "vcc", "scc", "memory"\n')
    out.write("#define PP_REGACC_KMAX 192\n")
    out.write("#define PP_REGACC_KDEEP 160\n")


if __name__ == "__main__":
    main()
