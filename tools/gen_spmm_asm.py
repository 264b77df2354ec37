#!/usr/bin/env python3
"""Generates climate_toolbox_amd/csrc/wagg_spmm_asm.inc: the per-(wave, chunk) inner loop of the
entry-list kernel (wagg_spmm.hip) as one inline-asm string, SPMM_CHUNK_ASM.

Why generated: the loop is unrolled over 5 blocks x 8 groups x 8 entries with static lane numbers
(v_readlane) and a two-stage software pipeline; writing 2,400 instructions by hand invites slips.

What one invocation does (one wave, one 256-cell chunk whose X tile is already in LDS):
  1. issues ALL entry loads of this chunk's list (5 blocks of 64 entries: lane j <- entry j, the
     8-byte entries stored as groups of [8 x lo][8 x weight]), then the 4 LDS-DMA pieces of the NEXT
     chunk's X tile and one warm-up load for the list two chunks ahead -- so every later wait on an
     entry block is a COUNTED vmcnt that leaves the DMA in flight;
  2. per entry: two v_readlane (entry -> SGPRs), v_bfi (LDS address), ds_read_b32 (64 timesteps of the
     cell), s_set_gpr_idx_idx + ONE v_fma_f32 into the accumulator register the entry names
     (v[40 + (lo & 0xff)]), group g+1's reads in flight while group g is accumulated;
  3. lists longer than 40 groups (never at 1 % fill) finish in a plain one-group-at-a-time loop.

Private registers (clobbered, hard-coded): v3 entry-load lane offset, v[4:8] / v[9:13] lo / weight
of blocks 0-4, v[14:21] / v[22:29] the two LDS-read sets, v30 lane*4 | buffer bit, v31 0xff00;
s[36:51] / s[52:67] the two entry sets, s68 saved M0, s[70:71] list pointer.  Accumulators v[40:127]
are in/out operands of the statement.
"""
import os

N_BLOCKS, GROUPS_PER_BLOCK = 5, 8
SA, SB = 36, 52          # SGPR sets
TA, TB = 14, 22          # VGPR temp sets
LO0, HI0 = 4, 9          # entry block registers
VO, LB, VMASK = 3, 30, 31
ACC0 = 40                # first accumulator register: v[40:127] = 88 accumulators
N_VM_AFTER = 4 + 1       # vector-memory ops issued after the entry loads: 4 DMA pieces + 1 warm-up


def issue(g, S, T, out):
    b, base = divmod(g, GROUPS_PER_BLOCK)
    base *= 8
    if base == 0:            # first group of a block: its two loads must have landed
        out.append("s_waitcnt vmcnt(%d)" % (2 * (N_BLOCKS - 1 - b) + N_VM_AFTER))
    for k in range(8):
        out.append("v_readlane_b32 s%d, v%d, %d" % (S + 2 * k, LO0 + b, base + k))
        out.append("v_readlane_b32 s%d, v%d, %d" % (S + 2 * k + 1, HI0 + b, base + k))
    for k in range(8):
        out.append("v_bfi_b32 v%d, v%d, s%d, v%d" % (T + k, VMASK, S + 2 * k, LB))
    for k in range(8):
        out.append("ds_read_b32 v%d, v%d" % (T + k, T + k))


def fma(S, T, younger, out):
    """accumulate one group; `younger` = LDS reads issued after this group's (next group's 8, or 0)"""
    for k in range(8):
        out.append(("s_set_gpr_idx_on s%d, 0xc" if k == 0 else "s_set_gpr_idx_idx s%d") % (S + 2 * k))
        out.append("s_waitcnt lgkmcnt(%d)" % (7 - k + younger))
        out.append("v_fma_f32 v%d, v%d, s%d, v%d" % (ACC0, T + k, S + 2 * k + 1, ACC0))
    out.append("s_set_gpr_idx_off")


def main():
    o = []
    o.append("s_mov_b32 s68, m0")
    o.append("s_mov_b32 s70, %[plo]")
    o.append("s_mov_b32 s71, %[phi]")
    # lane-derived registers
    o.append("v_mbcnt_lo_u32_b32 v14, -1, 0")
    o.append("v_mbcnt_hi_u32_b32 v14, -1, v14")              # v14 = lane
    o.append("v_lshrrev_b32 v15, 3, v14")
    o.append("v_and_b32 v16, 7, v14")
    o.append("v_lshlrev_b32 v15, 6, v15")
    o.append("v_lshl_or_b32 v%d, v16, 2, v15" % VO)          # (lane >> 3) * 64 + (lane & 7) * 4
    o.append("v_lshlrev_b32 v%d, 2, v14" % LB)
    o.append("v_or_b32 v%d, %%[bufbit], v%d" % (LB, LB))     # lane * 4 | buffer bit
    o.append("v_mov_b32 v%d, 0xff00" % VMASK)
    # 1. entry loads of this chunk's list
    for b in range(N_BLOCKS):
        o.append("global_load_dword v%d, v%d, s[70:71] offset:%d" % (LO0 + b, VO, 512 * b))
        o.append("global_load_dword v%d, v%d, s[70:71] offset:%d" % (HI0 + b, VO, 512 * b + 32))
    # the next chunk's X tile: 4 x 1 KiB LDS-DMA pieces of this wave
    o.append("v_lshlrev_b32 v15, 4, v14")                    # lane * 16
    for i in range(4):
        if i:
            o.append("v_add_u32 v16, 0x%x, v15" % (0x400 * i))
        o.append("s_add_u32 m0, %%[l0], 0x%x" % (0x400 * i) if i else "s_mov_b32 m0, %[l0]")
        o.append("s_nop 0")
        o.append("global_load_lds_dwordx4 v%d, %%[src]" % (16 if i else 15))
    # warm-up of the list two chunks ahead (into the sink; one 64-byte line per lane, clamped)
    o.append("v_min_u32 v17, %[wlim], v14")
    o.append("v_lshlrev_b32 v17, 6, v17")
    o.append("s_mov_b32 m0, %[sink]")
    o.append("s_nop 0")
    o.append("global_load_lds_dword v17, %[wsrc]")
    # 2. the pipelined groups
    o.append("s_cmp_eq_u32 %[n], 0")
    o.append("s_cbranch_scc1 8f")
    n_groups = N_BLOCKS * GROUPS_PER_BLOCK
    issue(0, SA, TA, o)
    for g in range(n_groups):
        S, T = (SA, TA) if g % 2 == 0 else (SB, TB)
        S2, T2 = (SB, TB) if g % 2 == 0 else (SA, TA)
        last = g == n_groups - 1
        if not last:
            issue(g + 1, S2, T2, o)
        fma(S, T, 0 if last else 8, o)
        o.append("s_sub_u32 %[n], %[n], 1")
        o.append("s_cmp_eq_u32 %[n], 0")
        o.append("s_cbranch_scc1 8f")
    # 3. overflow: one group at a time
    o.append("s_add_u32 s70, s70, %d" % (64 * n_groups))
    o.append("s_addc_u32 s71, s71, 0")
    o.append("7:")
    o.append("global_load_dword v%d, v%d, s[70:71] offset:0" % (LO0, VO))
    o.append("global_load_dword v%d, v%d, s[70:71] offset:32" % (HI0, VO))
    o.append("s_waitcnt vmcnt(0)")
    for k in range(8):
        o.append("v_readlane_b32 s%d, v%d, %d" % (SA + 2 * k, LO0, k))
        o.append("v_readlane_b32 s%d, v%d, %d" % (SA + 2 * k + 1, HI0, k))
    for k in range(8):
        o.append("v_bfi_b32 v%d, v%d, s%d, v%d" % (TA + k, VMASK, SA + 2 * k, LB))
    for k in range(8):
        o.append("ds_read_b32 v%d, v%d" % (TA + k, TA + k))
    fma(SA, TA, 0, o)
    o.append("s_add_u32 s70, s70, 64")
    o.append("s_addc_u32 s71, s71, 0")
    o.append("s_sub_u32 %[n], %[n], 1")
    o.append("s_cmp_eq_u32 %[n], 0")
    o.append("s_cbranch_scc0 7b")
    o.append("8:")
    o.append("s_waitcnt lgkmcnt(0)")          # reads issued for a group past the end are never consumed
    o.append("s_mov_b32 m0, s68")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "climate_toolbox_amd", "csrc",
                        "wagg_spmm_asm.inc")
    with open(path, "w") as f:
        f.write("// GENERATED by tools/gen_spmm_asm.py -- do not edit; see that script for the register map.\n")
        f.write("#define SPMM_CHUNK_ASM \\\n")
        for line in o:
            f.write('    "%s\\n\\t" \\\n' % line)
        f.write('    ""\n')
        clob = ["v%d" % i for i in range(3, 32)] + ["s%d" % i for i in range(36, 72)]
        f.write("#define SPMM_CHUNK_CLOBBERS %s\n" % ", ".join('"%s"' % c for c in clob))
    print("wrote", path, len(o), "instructions")


if __name__ == "__main__":
    main()
