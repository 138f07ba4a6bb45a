"""Desk-check of gemm_wg2.hip's LDS mapping (no GPU): every 16-byte piece a direct-to-LDS DMA instruction drops into a stage must be
the piece the fragment read of (row lane & 15, k chunk lane >> 4) expects there, and a ds_read_b128 must not collide inside any
group of 8 consecutive lanes (128 bytes per clock, 16-byte slots modulo the 256-byte bank row).   python tools/wg2_mapping_check.py"""
ROWB = 64
A_BYTES = 128 * ROWB

lds = {}
for op, ngroups, base in (("A", 8, 0), ("W", 16, A_BYTES)):
    for g in range(ngroups):
        for lane in range(64):
            drow = lane >> 2
            dch = (lane & 3) ^ ((drow >> 2) & 1)               # logical chunk this lane fetches (swizzle on the source address)
            addr = base + g * 1024 + lane * 16                 # the DMA writes lane-linear
            assert addr not in lds
            lds[addr] = (op, g * 16 + drow, dch)

bad = 0
for lane in range(64):
    l15, lq = lane & 15, lane >> 4
    xo = (lq ^ ((l15 >> 2) & 1)) << 4
    for h in range(2):
        for i in range(4):
            bad += lds.get(l15 * ROWB + xo + (h * 64 + i * 16) * ROWB) != ("A", h * 64 + i * 16 + l15, lq)
    for wave in range(4):
        for j in range(4):
            bad += lds.get(A_BYTES + (wave * 64 + l15) * ROWB + xo + j * 16 * ROWB) != ("W", wave * 64 + j * 16 + l15, lq)
worst = 0
for blk in range(8):
    for grp in range(8):
        slots = {(((lane & 15) * ROWB + (((lane >> 4) ^ (((lane & 15) >> 2) & 1)) << 4) + blk * 16 * ROWB) % 256) // 16
                 for lane in range(grp * 8, grp * 8 + 8)}
        worst = max(worst, 8 - len(slots))
print(f"fragment / DMA mismatches: {bad}; worst collisions inside an 8-lane group: {worst}")
raise SystemExit(1 if bad or worst else 0)
