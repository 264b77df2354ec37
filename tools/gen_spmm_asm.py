#!/usr/bin/env python3
"""Generates climate_toolbox_amd/csrc/wagg_spmm_asm.inc: the per-(wave, chunk) inner loop of the
entry-list kernel (wagg_spmm.hip) as inline-asm strings.

Why generated: the loop is unrolled over blocks x groups x entries with static lane numbers
(v_readlane) and a two-stage software pipeline; writing ~1,500 instructions by hand invites slips.

Geometry (must match wagg_dense_int.h): a wave's 64 lanes hold NT = 2 timesteps each (a 128-timestep
block), a chunk is KC = 128 grid cells x 512 B = 64 KiB of LDS, accumulators are register PAIRS
v[40 + 2j : 41 + 2j] for the wave's region j (j < 44; pair 43 swallows the padding entries).

SPMM_LOAD_LIST_ASM    (item prologue) loads the first chunk's list into register set A.
SPMM_CHUNK_ASM_A / _B one wave, one chunk whose X tile is already in LDS, the list of THIS chunk in set
                      A (B), in order:
  1. issues the loads of the NEXT chunk's list into the other set (consumed by the next statement: the
     C++ between two statements is scalar-only, which tools/check_spmm_codegen.py verifies on the
     compiled kernel) and the 4 LDS-DMA pieces of the next chunk's X tile -- all retired by the
     `s_waitcnt vmcnt(0)` + barrier that ends the chunk;
  2. per entry: two v_readlane (entry -> SGPRs), v_bfi (LDS address), ds_read_b64 (2 x 64 timesteps of
     the cell), s_set_gpr_idx_idx + v_pk_fma_f32 (or two v_fma_f32) into the accumulator pair the entry
     names (VGPR index mode), half-group h+1's reads in flight while half-group h is accumulated;
  3. lists longer than 16 groups (never at 1 % fill) finish in a one-group-at-a-time loop.

Private registers (clobbered, hard-coded; v3, v12, v30, v31 are set by the item prologue and live for the whole
item): v3 entry-load lane offset; set A = v[4:5] lo, v[6:7] weights
of blocks 0-1, set B = v[8:9], v[10:11]; v12 lane, v13 scratch; v[14:21] / v[22:29] the two LDS-read sets
(4 register pairs each); v30 lane*8 | buffer bit; v31 0xfe00; s[36:43] / s[44:51] the two half-group
entry sets; s68 saved M0; s[70:71] list pointer.
"""
import os
import sys

N_BLOCKS = 2                 # 64-entry blocks loaded per list (16 groups); longer lists take the slow loop
SA, SB = 36, 44              # SGPR half-group sets (4 entries x (lo, w))
TP, TQ = 14, 22              # VGPR temp sets: 4 pairs each
SETS = {"A": (4, 6), "B": (8, 10)}     # (first lo register, first weight register) of a list set
VO, LANE, SCR, LB, VMASK = 3, 12, 13, 30, 31
CUR_LB = LB                  # the register holding lane * 8 | buffer base for the statement being generated
ACC0 = 40
MASK = 0xfe00                # cell row bits of an entry: address = (lo & MASK) | lb
PK = int(os.environ.get("SPMM_PK", "1"))   # 1: v_pk_fma_f32, 0: two v_fma_f32
# ablation variants for tools/spmm_ablate.sh (timing only, results are wrong): any of nofma, nolds, noidx, now, nobfi,
# nodma, halfdma, nolist, noent; per-chunk overhead experiments (results stay right): nohoist, nonop
ABL = set(filter(None, os.environ.get("SPMM_ABL", "").split(",")))


def lane_regs(o, with_lb):
    o.append("v_mbcnt_lo_u32_b32 v%d, -1, 0" % LANE)
    o.append("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (LANE, LANE))
    o.append("v_lshrrev_b32 v%d, 3, v%d" % (SCR, LANE))
    o.append("v_and_b32 v%d, 7, v%d" % (VO, LANE))
    o.append("v_lshlrev_b32 v%d, 6, v%d" % (SCR, SCR))
    o.append("v_lshl_or_b32 v%d, v%d, 2, v%d" % (VO, VO, SCR))      # (lane >> 3) * 64 + (lane & 7) * 4
    if with_lb:
        o.append("v_lshlrev_b32 v%d, 3, v%d" % (LB, LANE))
        o.append("v_or_b32 v%d, %%[bufbit], v%d" % (LB, LB))        # lane * 8 | buffer bit
        o.append("v_mov_b32 v%d, 0x%x" % (VMASK, MASK))


def load_list(o, which):
    lo0, hi0 = SETS[which]
    for b in range(N_BLOCKS):
        o.append("global_load_dword v%d, v%d, s[70:71] offset:%d" % (lo0 + b, VO, 512 * b))
        o.append("global_load_dword v%d, v%d, s[70:71] offset:%d" % (hi0 + b, VO, 512 * b + 32))


def issue(h, which, S, T, o):
    """half-group h (4 entries): entry -> SGPRs, LDS address, read"""
    lo0, hi0 = SETS[which]
    b, base = divmod(4 * h, 64)
    for k in range(4):
        o.append("v_readlane_b32 s%d, v%d, %d" % (S + 2 * k, lo0 + b, base + k))
        if "now" not in ABL:
            o.append("v_readlane_b32 s%d, v%d, %d" % (S + 2 * k + 1, hi0 + b, base + k))
    for k in range(4):
        if "nobfi" not in ABL:
            o.append("v_bfi_b32 v%d, v%d, s%d, v%d" % (T + 2 * k, VMASK, S + 2 * k, CUR_LB))
    for k in range(4):
        if "nolds" not in ABL:
            o.append("ds_read_b64 v[%d:%d], v%d" % (T + 2 * k, T + 2 * k + 1, T + 2 * k))


def fma(S, T, younger, o):
    for k in range(4):
        if "noidx" not in ABL:
            o.append(("s_set_gpr_idx_on s%d, 0xc" if k == 0 else "s_set_gpr_idx_idx s%d") % (S + 2 * k))
        if "nolds" not in ABL:
            o.append("s_waitcnt lgkmcnt(%d)" % (3 - k + younger))
        if "nofma" in ABL:
            continue
        if PK:
            o.append("v_pk_fma_f32 v[%d:%d], v[%d:%d], s[%d:%d], v[%d:%d] op_sel:[0,1,0]"
                     % (ACC0, ACC0 + 1, T + 2 * k, T + 2 * k + 1, S + 2 * k, S + 2 * k + 1, ACC0, ACC0 + 1))
        else:
            o.append("v_fma_f32 v%d, v%d, s%d, v%d" % (ACC0, T + 2 * k, S + 2 * k + 1, ACC0))
            o.append("v_fma_f32 v%d, v%d, s%d, v%d" % (ACC0 + 1, T + 2 * k + 1, S + 2 * k + 1, ACC0 + 1))
    if "noidx" not in ABL:
        o.append("s_set_gpr_idx_off")


def chunk(cur, nxt):
    global CUR_LB
    o = []
    o.append("s_mov_b32 s68, m0")
    o.append("s_mov_b32 s70, %[nplo]")
    o.append("s_mov_b32 s71, %[nphi]")
    hoist, nonop = "nohoist" not in ABL, "nonop" in ABL
    if hoist:
        CUR_LB = LB                                            # v[LANE], v[VO], v[VMASK], v[LB] come from the item prologue
    else:
        CUR_LB = LB
        lane_regs(o, True)
    if "nolist" not in ABL:
        load_list(o, nxt)                                      # 1. the NEXT chunk's list
    n_dma = 2 if "halfdma" in ABL else 4
    if nonop:       # the vector instruction between the M0 write and the LDS-DMA is the wait state the pair needs
        for i in range(n_dma):
            o.append("s_add_u32 m0, %%[l0], 0x%x" % (0x400 * i) if i else "s_mov_b32 m0, %[l0]")
            if i:
                o.append("v_add_u32 v%d, 0x%x, v%d" % (TP + 1, 0x400 * i, TP))
            else:
                o.append("v_lshlrev_b32 v%d, 4, v%d" % (TP, LANE))     # lane * 16: LDS-DMA of the next X tile
            if "nodma" not in ABL:
                o.append("global_load_lds_dwordx4 v%d, %%[src]" % (TP + 1 if i else TP))
    else:
        o.append("v_lshlrev_b32 v%d, 4, v%d" % (TP, LANE))     # lane * 16: LDS-DMA of the next X tile
        for i in range(n_dma):
            if i:
                o.append("v_add_u32 v%d, 0x%x, v%d" % (TP + 1, 0x400 * i, TP))
            o.append("s_add_u32 m0, %%[l0], 0x%x" % (0x400 * i) if i else "s_mov_b32 m0, %[l0]")
            o.append("s_nop 0")
            if "nodma" not in ABL:
                o.append("global_load_lds_dwordx4 v%d, %%[src]" % (TP + 1 if i else TP))
    # (an L2 warm-up load of the list three chunks ahead used to sit here: measured 2.1 ms SLOWER on the c5 rank
    #  shard -- profiles/r02_spmm_ablation.txt -- and removed together with its operands)
    if hoist and cur == "B":                                   # statement B reads LDS buffer 1
        o.append("v_add_u32 v%d, 0x10000, v%d" % (SCR, LB))
        CUR_LB = SCR
    o.append("s_cmp_eq_u32 %[n], 0")                           # 2. the entries of THIS chunk
    o.append("s_cbranch_scc1 8f")
    if "noent" in ABL:
        o.append("s_branch 8f")
    n_half = N_BLOCKS * 16
    issue(0, cur, SA, TP, o)
    for h in range(n_half):
        S, T = (SA, TP) if h % 2 == 0 else (SB, TQ)
        S2, T2 = (SB, TQ) if h % 2 == 0 else (SA, TP)
        last = h == n_half - 1
        if not last:
            issue(h + 1, cur, S2, T2, o)
        fma(S, T, 0 if last else 4, o)
        if h % 2 == 1:                                         # a whole group done
            o.append("s_sub_u32 %[n], %[n], 1")
            o.append("s_cmp_eq_u32 %[n], 0")
            o.append("s_cbranch_scc1 8f")
    # 3. overflow: one group at a time from behind the loaded blocks (uses the current set's registers)
    lo0, hi0 = SETS[cur]
    o.append("s_add_u32 s70, %%[cplo], %d" % (512 * N_BLOCKS))
    o.append("s_addc_u32 s71, %[cphi], 0")
    o.append("7:")
    o.append("global_load_dword v%d, v%d, s[70:71] offset:0" % (lo0, VO))
    o.append("global_load_dword v%d, v%d, s[70:71] offset:32" % (hi0, VO))
    o.append("s_waitcnt vmcnt(0)")
    for hh in range(2):
        for k in range(4):
            o.append("v_readlane_b32 s%d, v%d, %d" % (SA + 2 * k, lo0, 4 * hh + k))
            o.append("v_readlane_b32 s%d, v%d, %d" % (SA + 2 * k + 1, hi0, 4 * hh + k))
        for k in range(4):
            o.append("v_bfi_b32 v%d, v%d, s%d, v%d" % (TP + 2 * k, VMASK, SA + 2 * k, CUR_LB))
        for k in range(4):
            o.append("ds_read_b64 v[%d:%d], v%d" % (TP + 2 * k, TP + 2 * k + 1, TP + 2 * k))
        fma(SA, TP, 0, o)
    o.append("s_add_u32 s70, s70, 64")
    o.append("s_addc_u32 s71, s71, 0")
    o.append("s_sub_u32 %[n], %[n], 1")
    o.append("s_cmp_eq_u32 %[n], 0")
    o.append("s_cbranch_scc0 7b")
    o.append("8:")
    o.append("s_waitcnt lgkmcnt(0)")          # reads issued for a half-group past the end are never consumed
    o.append("s_mov_b32 m0, s68")
    return o


def prologue():
    o = []
    o.append("s_mov_b32 s70, %[nplo]")
    o.append("s_mov_b32 s71, %[nphi]")
    lane_regs(o, "nohoist" not in ABL)  # lane, lane * 8 | base of LDS buffer 0 and the cell mask live for the whole item
    load_list(o, "A")
    return o


def emit(f, name, lines):
    f.write("#define %s \\\n" % name)
    for line in lines:
        f.write('    "%s\\n\\t" \\\n' % line)
    f.write('    ""\n')


def main():
    path = os.environ.get("SPMM_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                     "climate_toolbox_amd", "csrc", "wagg_spmm_asm.inc")
    a, b, p = chunk("A", "B"), chunk("B", "A"), prologue()
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_spmm_asm.py (SPMM_PK=%d%s) -- do not edit; see that script for the register map.\n"
                % (PK, " SPMM_ABL=" + ",".join(sorted(ABL)) if ABL else ""))
        emit(f, "SPMM_LOAD_LIST_ASM", p)
        emit(f, "SPMM_CHUNK_ASM_A", a)
        emit(f, "SPMM_CHUNK_ASM_B", b)
        clob = ["v%d" % i for i in range(3, 32)] + ["s%d" % i for i in range(36, 72)]
        f.write("#define SPMM_CHUNK_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob))
    print("wrote", path, len(a), "instructions per chunk statement", file=sys.stderr)


if __name__ == "__main__":
    main()
