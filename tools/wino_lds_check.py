"""Exhaustive bank check of the Winograd kernel's LDS image (csrc/wino.hip): rows of 16 floats, chunk c of row r in slot c ^ f(r).
ds_read_b128 is serviced in 4 groups of 16 lanes (MI355X_MICROARCH.md, LDS table); a group is conflict-free when its 16 lanes hit 16
distinct 16-byte slots of the 256-byte bank row.  ds_write_b64 groups: 4 x 16 contiguous lanes, 32 distinct 8-byte slots needed... 16 lanes x 8 B."""
F = [0, 2, 3, 1]
f = lambda r: F[(r >> 2) & 3]
groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
worst = 1
for base in range(0, 1024, 16):                      # any 16-row-aligned block of rows
    for grp in groups:
        slots = {}
        for lane in grp:
            l16, g = lane & 15, lane >> 4
            row = base + l16
            addr16 = row * 4 + (g ^ f(row))          # address in 16-byte units
            slots.setdefault(addr16 % 16, set()).add(addr16)
        worst = max(worst, max(len(v) for v in slots.values()))
print("ds_read_b128 A fragments: worst conflict degree", worst)
assert worst == 1
# ds_write_b128 of the transform: producer lane = (tile & 15) * 4 + cq within a wave (ptid = tile * 4 + cq); groups of 8 contiguous lanes
worst = 1
for pos in range(16):
    for wave in range(4):
        for grp in range(8):
            slots = {}
            for lane in range(grp * 8, grp * 8 + 8):
                ptid = wave * 64 + lane
                tl, cq = ptid >> 2, ptid & 3
                addr16 = (pos * 64 + tl) * 4 + (cq ^ f(tl))
                slots.setdefault(addr16 % 16, set()).add(addr16)
            worst = max(worst, max(len(v) for v in slots.values()))
print("ds_write_b128 V stores: worst conflict degree", worst)
assert worst == 1
