#!/usr/bin/env python3
"""Generates climate_toolbox_amd/csrc/wagg_spmm_asm.inc: the per-(wave, chunk) inner loop of the
entry-list kernel (wagg_spmm.hip) as inline-asm strings, once for fp32 and once for fp64.

Why generated: the loop is unrolled over blocks x groups x entries with static lane numbers
(v_readlane) and a two-stage software pipeline; writing ~1,500 instructions by hand invites slips.

Geometry (must match wagg_dense_int.h).  A chunk is KC = 128 grid cells x 512 B = 64 KiB of LDS; a cell's
row holds one time block: 128 fp32 timesteps (two per lane, ds_read_b64 at lane * 8 = timesteps 2 lane,
2 lane + 1) or 64 fp64 timesteps (one per lane).  Accumulators are register PAIRS v[40 + 2j : 41 + 2j]
for the wave's region j (j < 44; pair 43 swallows the padding entries): two fp32 timesteps, or one double.

Entry lists (round 3 format).  A group of 8 entries is stored as
    [4 x (lo16 of entry 2p | lo16 of entry 2p + 1 << 16)] [8 x weight]            fp32: 48 bytes
    [4 x lo16 pairs] [8 x weight (double)]                                        fp64: 80 bytes
with lo16 = cell_in_chunk << 9 | accumulator register offset (2 j).  One v_readlane brings the lo16 of TWO
entries into an SGPR (the odd entry takes `s_lshr_b32 ..., 16`, a scalar instruction).  The WEIGHTS of a list never
pass through SGPRs: the 16 groups a wave loads per chunk go to the wave's own LDS slots by LDS-DMA (one
global_load_lds_dwordx4: fp32 lanes 0-31, 16 bytes = four weights each; fp64 all lanes, two doubles each) together
with the next chunk's X tile, and come back as BROADCAST reads (every lane the same address, conflict-free): one
ds_read_b128 per half-group of four entries in fp32 (two in fp64).  So an entry costs
    0.5 v_readlane + v_bfi (LDS address of its cell) + ds_read_b64 (X) + 0.25 ds_read_b128 (weights; fp64 0.5)
    + s_set_gpr_idx_idx + ONE v_pk_fma_f32 (weight = half of a VGPR pair picked by op_sel) / v_fma_f64
whose accumulator pair the entry itself names through the VGPR index mode; one s_waitcnt per half-group.
(Round 2: [8 x lo32][8 x weight], weights through v_readlane: two v_readlane and a wait per entry, 53.2 ms on
c5-uniform; this format 46.4 ms -- profiles/r03_spmm_ablation_a.txt / _b.txt.)  `SPMM_ABL=wreg` generates the
intermediate form with the weights in registers + v_readlane (ablation only).

SPMM_LOAD_LIST_ASM_*     (item prologue) per-lane constants + the first chunk's list (lo16 pairs -> set A, weights ->
                         LDS weight buffer 0).
SPMM_CHUNK_ASM_A_* / _B_*  one wave, one chunk whose X tile is already in LDS, the list of THIS chunk in
                         set A (B) / weight buffer 0 (1), in order:
  1. issues the loads of the NEXT chunk's list (pairs into the other set, weights into the other weight buffer;
     consumed by the next statement: the C++ between two statements is scalar-only, which
     tools/check_spmm_codegen.py verifies on the compiled kernels) and the 4 LDS-DMA pieces of the next chunk's X
     tile -- all retired by the `s_waitcnt vmcnt(0)` + barrier that ends the chunk;
  2. the entries, half-group h + 1's reads in flight while half-group h is accumulated (fp64 has ONE weight register
     set: the weights of h + 1 are read behind the FMAs of h);
  3. lists longer than 16 groups (never at 1 % fill; a 9.5 % table has them) finish in a one-group-at-a-time loop
     with the weights through v_readlane.

LDS: [2 x 64 KiB X tiles][2 buffers x 16 waves x 128 weights] (SPMM_W_LDS0 = 0x20000; 144 KiB fp32, 160 KiB fp64).

Private registers (clobbered, hard-coded; class Geo holds the map).  Set by the item prologue and live for the whole
item: v3 lane offset of the lo16-pair loads, v4 lane offset of the weight DMA, v7 lane, v9 lane * 8 | buffer bit, v10
this wave's weight slots in LDS, v11 the cell mask 0xfe00.  v5 / v6 the lo16 pairs of list set A / B (128 entries
each); v8 scratch; v[12:19] / v[20:27] the two X read sets (4 register pairs each); v[28:31] / v[32:35] the weights of
the two half-group sets (fp64: one set v[28:35]).  SGPRs: lo16 of the two half-group sets s36.. / s44.. (fp64 s36.. /
s48..); s[62:63] saved exec; s68 saved M0; s[70:71] list pointer.
"""
import os
import sys

N_BLOCKS = 2                 # 64-entry blocks loaded per list (16 groups); longer lists take the slow loop
VPO, VWO = 3, 4              # lane offsets of the pair / weight loads
ACC0 = 40
MASK = 0xfe00                # cell row bits of an entry: address = (lo & MASK) | lb
V_LAST = 35
W_LDS0 = 0x20000             # weights in LDS, behind the two X buffers: [2 buffers][16 waves][128 weights]
# ablation variants for tools/spmm_ablate.sh (timing only, results are wrong): any of nofma, nolds, noidx, now, nobfi,
# nodma, halfdma, nolist, noent
ABL = set(filter(None, os.environ.get("SPMM_ABL", "").split(",")))


class Geo:
    def __init__(self, f64):
        self.f64 = f64
        self.sfx = "F64" if f64 else "F32"
        self.gwb = 80 if f64 else 48            # bytes per 8-entry group
        self.wlds = "wreg" not in ABL           # the weights go to LDS by LDS-DMA and come back by broadcast reads
        self.wb = 8 if f64 else 4               # bytes per weight
        self.wbuf_bytes = 16 * 128 * self.wb    # one weight buffer: 16 waves x 128 weights
        # SGPR half-group sets
        if f64:
            self.sets = {0: dict(lo=36, w=40), 1: dict(lo=48, w=52)}
        else:
            self.sets = {0: dict(lo=36, w=37), 1: dict(lo=44, w=45)}
        if self.wlds:
            # v5 / v6 the lo16 pairs of list set A / B; fp32: v[28:31] / v[32:35] the weights of the two half-group sets;
            # fp64: ONE set v[28:35] (four doubles), read behind the FMAs of the half-group before (see chunk)
            self.SETS = {"A": 5, "B": 6}
            self.LANE, self.SCR, self.LB, self.VWB, self.VMASK = 7, 8, 9, 10, 11
            self.TP, self.TQ = 12, 20
            self.WREG = {0: 28, 1: 28} if f64 else {0: 28, 1: 32}
        else:
            self.SETS = {"A": 5, "B": 10}       # +0 pairs, +1,+2 w (lo) of blocks 0,1, +3,+4 w hi (fp64)
            self.LANE, self.SCR, self.LB, self.VMASK = 15, 16, 17, 34
            self.TP, self.TQ = 18, 26

    def s_lo(self, s, k):
        return self.sets[s]["lo"] + (k if self.f64 else 2 * k)

    def s_w(self, s, k):
        return self.sets[s]["w"] + 2 * k


def lane_regs(g, o):
    LANE, SCR, LB = g.LANE, g.SCR, g.LB
    o.append("v_mbcnt_lo_u32_b32 v%d, -1, 0" % LANE)
    o.append("v_mbcnt_hi_u32_b32 v%d, -1, v%d" % (LANE, LANE))
    # pair loads: lane l <- pair (l & 3) of group (l >> 2): 128 entries in one register
    o.append("v_lshrrev_b32 v%d, 2, v%d" % (SCR, LANE))
    o.append("v_mul_u32_u24 v%d, %d, v%d" % (SCR, g.gwb, SCR))
    o.append("v_and_b32 v%d, 3, v%d" % (VPO, LANE))
    o.append("v_lshl_add_u32 v%d, v%d, 2, v%d" % (VPO, VPO, SCR))
    # weight loads: lane l <- weight (l & 7) of group (l >> 3) (+ 8 groups per block)
    o.append("v_lshrrev_b32 v%d, 3, v%d" % (SCR, LANE))
    o.append("v_mul_u32_u24 v%d, %d, v%d" % (SCR, g.gwb, SCR))
    o.append("v_and_b32 v%d, 7, v%d" % (VWO, LANE))
    o.append("v_lshl_add_u32 v%d, v%d, 2, v%d" % (VWO, VWO, SCR))      # (an add: 48 k and (lane & 7) * 4 share bit 4)
    if g.wlds:
        # weights by ONE LDS-DMA of 16 bytes per lane, lanes 0-31 (exec-masked): lane l <- weights 4 l .. 4 l + 3 =
        # half (l & 1) of group (l >> 1); the 16 bytes of the pairs in front are part of the register (an LDS-DMA's
        # immediate offset would move its LDS destination as well)
        # (fp64: 16 bytes = 2 doubles: lane l <- quarter (l & 3) of group (l >> 2), all 64 lanes)
        sh, m = (2, 3) if g.f64 else (1, 1)
        o.append("v_lshrrev_b32 v%d, %d, v%d" % (SCR, sh, LANE))
        o.append("v_mul_u32_u24 v%d, %d, v%d" % (SCR, g.gwb, SCR))
        o.append("v_and_b32 v%d, %d, v%d" % (VWO, m, LANE))
        o.append("v_lshl_add_u32 v%d, v%d, 4, v%d" % (VWO, VWO, SCR))
        o.append("v_add_u32 v%d, 16, v%d" % (VWO, VWO))
    o.append("v_lshlrev_b32 v%d, 3, v%d" % (LB, LANE))
    o.append("v_or_b32 v%d, %%[bufbit], v%d" % (LB, LB))            # lane * 8 | buffer bit
    o.append("v_mov_b32 v%d, 0x%x" % (g.VMASK, MASK))
    if g.wlds:
        o.append("v_mov_b32 v%d, %%[wbase]" % g.VWB)                # this wave's weight slots in LDS (buffer 0)


def load_list(g, o, which, ptr="s[70:71]"):
    """the list at `ptr`: lo16 pairs -> set `which`; weights -> registers, or (fp32) -> this wave's LDS slots at
    %[wl0] by LDS-DMA (lane l <- weight of entry l: the same per-lane source offsets)"""
    r0 = g.SETS[which]
    o.append("global_load_dword v%d, v%d, %s" % (r0, VPO, ptr))
    if g.wlds and g.f64:
        o.append("s_mov_b32 m0, %[wl0]")
        o.append("s_nop 0")
        o.append("global_load_lds_dwordx4 v%d, %s" % (VWO, ptr))
        return
    if g.wlds:
        o.append("s_mov_b64 s[62:63], exec")
        o.append("s_mov_b32 m0, %[wl0]")
        o.append("s_mov_b64 exec, 0xffffffff")                 # lanes 0-31 (also the wait state behind the M0 write)
        o.append("global_load_lds_dwordx4 v%d, %s" % (VWO, ptr))
        o.append("s_mov_b64 exec, s[62:63]")
        return
    for b in range(N_BLOCKS):
        o.append("global_load_dword v%d, v%d, %s offset:%d" % (r0 + 1 + b, VWO, ptr, 8 * g.gwb * b + 16))
        if g.f64:
            o.append("global_load_dword v%d, v%d, %s offset:%d" % (r0 + 3 + b, VWO, ptr, 8 * g.gwb * b + 48))


def wread(g, h, s, o, wbuf):
    """the four weights of half-group h: broadcast reads (every lane the same address)"""
    if not g.wlds or "now" in ABL:
        return
    W = g.WREG[s]
    at = wbuf * g.wbuf_bytes + 4 * g.wb * h
    o.append("ds_read_b128 v[%d:%d], v%d offset:%d" % (W, W + 3, g.VWB, at))
    if g.f64:
        o.append("ds_read_b128 v[%d:%d], v%d offset:%d" % (W + 4, W + 7, g.VWB, at + 16))


def issue(g, h, which, s, T, o, cur_lb, wbuf, with_w=True):
    """half-group h (entries 4h .. 4h + 3 of the loaded blocks): entries -> SGPRs, LDS addresses, reads"""
    r0 = g.SETS[which]
    for k in range(4):
        e = 4 * h + k
        if k % 2 == 0:
            o.append("v_readlane_b32 s%d, v%d, %d" % (g.s_lo(s, k), r0, e >> 1))
        else:
            o.append("s_lshr_b32 s%d, s%d, 16" % (g.s_lo(s, k), g.s_lo(s, k - 1)))
        if "now" not in ABL and not g.wlds:
            b, l = divmod(e, 64)
            if g.f64:
                o.append("v_readlane_b32 s%d, v%d, %d" % (g.s_w(s, k), r0 + 1 + b, l))
                o.append("v_readlane_b32 s%d, v%d, %d" % (g.s_w(s, k) + 1, r0 + 3 + b, l))
            else:
                o.append("v_readlane_b32 s%d, v%d, %d" % (g.s_w(s, k), r0 + 1 + b, l))
    for k in range(4):
        if "nobfi" not in ABL:
            o.append("v_bfi_b32 v%d, v%d, s%d, v%d" % (T + 2 * k, g.VMASK, g.s_lo(s, k), cur_lb))
    for k in range(4):
        if "nolds" not in ABL:
            o.append("ds_read_b64 v[%d:%d], v%d" % (T + 2 * k, T + 2 * k + 1, T + 2 * k))
    if with_w:
        wread(g, h, s, o, wbuf)


def x_ops(g):
    return 0 if "nolds" in ABL else 4


def w_ops(g):
    return ((2 if g.f64 else 1) if g.wlds and "now" not in ABL else 0)


def lds_ops(g):
    """LDS instructions per half-group"""
    return x_ops(g) + w_ops(g)


def fma(g, s, T, younger, o, sgpr_w=False):
    """accumulate a half-group; `younger` = LDS reads issued after this half-group's own"""
    if lds_ops(g):
        o.append("s_waitcnt lgkmcnt(%d)" % younger)
    for k in range(4):
        if "noidx" not in ABL:
            o.append(("s_set_gpr_idx_on s%d, 0xc" if k == 0 else "s_set_gpr_idx_idx s%d") % g.s_lo(s, k))
        if "nofma" in ABL:
            continue
        x = "v[%d:%d]" % (T + 2 * k, T + 2 * k + 1)
        acc = "v[%d:%d]" % (ACC0, ACC0 + 1)
        if g.f64 and g.wlds and not sgpr_w:
            W = g.WREG[s] + 2 * k
            o.append("v_fma_f64 %s, %s, v[%d:%d], %s" % (acc, x, W, W + 1, acc))
        elif g.f64:
            o.append("v_fma_f64 %s, %s, s[%d:%d], %s" % (acc, x, g.s_w(s, k), g.s_w(s, k) + 1, acc))
        elif g.wlds and not sgpr_w:
            W = g.WREG[s] + 2 * (k // 2)      # weights k, k + 1 sit in one aligned register pair
            sel = "op_sel:[0,1,0]" if k % 2 else "op_sel:[0,0,0] op_sel_hi:[1,0,1]"
            o.append("v_pk_fma_f32 %s, %s, v[%d:%d], %s %s" % (acc, x, W, W + 1, acc, sel))
        else:
            o.append("v_pk_fma_f32 %s, %s, s[%d:%d], %s op_sel:[0,1,0]" % (acc, x, g.s_lo(s, k), g.s_lo(s, k) + 1, acc))
    if "noidx" not in ABL:
        o.append("s_set_gpr_idx_off")


def chunk(g, cur, nxt):
    o = []
    TP, TQ, LANE, SCR, LB = g.TP, g.TQ, g.LANE, g.SCR, g.LB
    o.append("s_mov_b32 s68, m0")
    o.append("s_mov_b32 s70, %[nplo]")
    o.append("s_mov_b32 s71, %[nphi]")
    if "nolist" not in ABL:
        load_list(g, o, nxt)                                   # 1. the NEXT chunk's list
    n_dma = 2 if "halfdma" in ABL else 4
    o.append("v_lshlrev_b32 v%d, 4, v%d" % (TP, LANE))         # lane * 16: LDS-DMA of the next X tile
    for i in range(n_dma):
        if i:
            o.append("v_add_u32 v%d, 0x%x, v%d" % (TP + 1, 0x400 * i, TP))
        o.append("s_add_u32 m0, %%[l0], 0x%x" % (0x400 * i) if i else "s_mov_b32 m0, %[l0]")
        o.append("s_nop 0")
        if "nodma" not in ABL:
            o.append("global_load_lds_dwordx4 v%d, %%[src]" % (TP + 1 if i else TP))
    cur_lb = LB
    wbuf = 0
    if cur == "B":                                             # statement B reads X buffer 1 / weight buffer 1
        o.append("v_add_u32 v%d, 0x10000, v%d" % (SCR, LB))
        cur_lb = SCR
        wbuf = 1
    o.append("s_sub_u32 %[n], %[n], 1")                        # 2. the entries of THIS chunk; n - 1 groups remain behind
    o.append("s_cbranch_scc1 8f")                              #    the first (a borrow: the list is empty)
    if "noent" in ABL:
        o.append("s_branch 8f")
    n_half = N_BLOCKS * 16
    one_wset = g.wlds and g.WREG[0] == g.WREG[1]               # (fp64) the weights of h + 1 are read behind the FMAs of h
    issue(g, 0, cur, 0, TP, o, cur_lb, wbuf)
    for h in range(n_half):
        s, T = (0, TP) if h % 2 == 0 else (1, TQ)
        s2, T2 = (1, TQ) if h % 2 == 0 else (0, TP)
        last = h == n_half - 1
        if not last:
            issue(g, h + 1, cur, s2, T2, o, cur_lb, wbuf, with_w=not one_wset)
        # LDS reads in flight behind this half-group's own: the X reads of h + 1 (and its weight reads unless they
        # come later)
        fma(g, s, T, 0 if last else (x_ops(g) if one_wset else lds_ops(g)), o)
        if one_wset and not last:
            wread(g, h + 1, s2, o, wbuf)
        if h % 2 == 1:                                         # a whole group done: was it the last one?
            o.append("s_sub_u32 %[n], %[n], 1")
            o.append("s_cbranch_scc1 8f")
    # 3. overflow: one group at a time from behind the loaded blocks: its lo16 pairs into lanes 0-3 of the current
    #    set's pair register, its weights into lanes 0-7 of temporaries (TQ), entries one half-group at a time
    r0 = g.SETS[cur]
    wr = TQ if g.wlds else r0 + 1
    wh = TQ + 1 if g.wlds else r0 + 3
    o.append("s_add_u32 s70, %%[cplo], %d" % (8 * g.gwb * N_BLOCKS))
    o.append("s_addc_u32 s71, %[cphi], 0")
    o.append("7:")
    o.append("global_load_dword v%d, v%d, s[70:71]" % (r0, VPO))
    if g.wlds:
        o.append("v_lshlrev_b32 v%d, %d, v%d" % (TQ + 2, 3 if g.f64 else 2, LANE))   # lane e <- weight e of the group
        o.append("global_load_dword v%d, v%d, s[70:71] offset:16" % (wr, TQ + 2))
        if g.f64:
            o.append("global_load_dword v%d, v%d, s[70:71] offset:20" % (wh, TQ + 2))
    else:
        o.append("global_load_dword v%d, v%d, s[70:71] offset:16" % (wr, VWO))
        if g.f64:
            o.append("global_load_dword v%d, v%d, s[70:71] offset:48" % (wh, VWO))
    o.append("s_waitcnt vmcnt(0)")
    for hh in range(2):
        for k in range(4):
            e = 4 * hh + k
            if k % 2 == 0:
                o.append("v_readlane_b32 s%d, v%d, %d" % (g.s_lo(0, k), r0, e >> 1))
            else:
                o.append("s_lshr_b32 s%d, s%d, 16" % (g.s_lo(0, k), g.s_lo(0, k - 1)))
            o.append("v_readlane_b32 s%d, v%d, %d" % (g.s_w(0, k), wr, e))
            if g.f64:
                o.append("v_readlane_b32 s%d, v%d, %d" % (g.s_w(0, k) + 1, wh, e))
        for k in range(4):
            o.append("v_bfi_b32 v%d, v%d, s%d, v%d" % (TP + 2 * k, g.VMASK, g.s_lo(0, k), cur_lb))
        for k in range(4):
            o.append("ds_read_b64 v[%d:%d], v%d" % (TP + 2 * k, TP + 2 * k + 1, TP + 2 * k))
        o.append("s_waitcnt lgkmcnt(0)")
        saved = set(ABL)
        ABL.discard("nolds"); ABL.discard("nofma"); ABL.discard("noidx")
        body = []
        fma(g, 0, TP, 0, body, sgpr_w=True)
        o.extend(b for b in body if not b.startswith("s_waitcnt"))
        ABL.clear(); ABL.update(saved)
    o.append("s_add_u32 s70, s70, %d" % g.gwb)
    o.append("s_addc_u32 s71, s71, 0")
    o.append("s_sub_u32 %[n], %[n], 1")
    o.append("s_cbranch_scc0 7b")
    o.append("8:")
    o.append("s_waitcnt lgkmcnt(0)")          # reads issued for a half-group past the end are never consumed
    o.append("s_mov_b32 m0, s68")
    return o


def prologue(g):
    o = []
    o.append("s_mov_b32 s68, m0")
    o.append("s_mov_b32 s70, %[nplo]")
    o.append("s_mov_b32 s71, %[nphi]")
    lane_regs(g, o)   # lane, load offsets, lane * 8 | base of LDS buffer 0 and the cell mask live for the whole item
    load_list(g, o, "A")
    o.append("s_mov_b32 m0, s68")
    return o


def emit(f, name, lines):
    f.write("#define %s \\\n" % name)
    for line in lines:
        f.write('    "%s\\n\\t" \\\n' % line)
    f.write('    ""\n')


def main():
    path = os.environ.get("SPMM_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                     "climate_toolbox_amd", "csrc", "wagg_spmm_asm.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_spmm_asm.py%s -- do not edit; see that script for the register map.\n"
                % (" SPMM_ABL=" + ",".join(sorted(ABL)) if ABL else ""))
        for f64 in (False, True):
            g = Geo(f64)
            a, b, p = chunk(g, "A", "B"), chunk(g, "B", "A"), prologue(g)
            emit(f, "SPMM_LOAD_LIST_ASM_" + g.sfx, p)
            emit(f, "SPMM_CHUNK_ASM_A_" + g.sfx, a)
            emit(f, "SPMM_CHUNK_ASM_B_" + g.sfx, b)
            print("wrote", path, g.sfx, len(a), "instructions per chunk statement", file=sys.stderr)
        clob = ["v%d" % i for i in range(3, V_LAST + 1)] + ["s%d" % i for i in range(36, 72)]
        f.write("#define SPMM_CHUNK_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob))
        f.write("#define SPMM_W_LDS0 0x%x\n" % W_LDS0)


if __name__ == "__main__":
    main()
