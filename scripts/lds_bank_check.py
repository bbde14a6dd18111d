"""Brute-force check of the LDS layouts of conv_c4_kernel (unet_c4.hip) against the ds_read_b128 bank model of
MI355X_MICROARCH.md: a wave64 ds_read_b128 is served in 4 groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, the same
+32); a group is conflict-free when its 16 lanes touch 16 different 16-byte slots of the 256-byte bank row (64 banks x 4 B).

  halo image   : pixel p at p * 64, piece c = plane * 2 + kgroup at position c ^ ((p >> 2) & 3); lane l reads pixel
                 base + (l & 31) (any base: tap shifts, rows, flattened offsets), kgroup l >> 5
  weight image : row n at n * 32, piece g at position g ^ ((n >> 3) & 1); lane l reads row 32 j + (l & 31), kgroup l >> 5
  epilogue window: pixel p at p * 128, piece g (4 channels, fp32) at position g ^ (((p & 1) << 2) | ((p >> 1) & 3)); written by
                 ds_write_b128 (8 groups of 8 contiguous lanes over the 8 slots of a 128-byte bank row): lane l = pixel l & 31,
                 piece 2 q + (l >> 5); read back by ds_read_b128: task t = pixel 16 t + (l >> 2), pieces 2 (l & 3) + h; pooling
                 reads: pixel 2 (l >> 2) + (d ^ ((l >> 4) & 1))
"""
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS += [[l + 32 for l in g] for g in GROUPS]


def conflicts(addr_of_lane):
    worst = 1
    for g in GROUPS:
        slots = {}
        for l in g:
            a = addr_of_lane(l)
            assert a % 16 == 0
            slots.setdefault((a // 16) % 16, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def halo_addr(base, plane):
    def f(l):
        p, g = base + (l & 31), l >> 5
        c = plane * 2 + g
        return p * 64 + ((c ^ ((p >> 2) & 3)) * 16)
    return f


def weight_addr(j, plane, slot_bytes):
    def f(l):
        n, g = 32 * j + (l & 31), l >> 5
        return plane * (slot_bytes // 2) + n * 32 + ((g ^ ((n >> 3) & 1)) * 16)
    return f


worst_h = max(conflicts(halo_addr(b, pl)) for b in range(0, 700) for pl in (0, 1))
worst_w = max(conflicts(weight_addr(j, pl, 8192)) for j in range(4) for pl in (0, 1))
print(f"halo image: worst {worst_h}-way over all bases 0..699 and both planes; weight image: worst {worst_w}-way")
assert worst_h == 1 and worst_w == 1


def write_conflicts(addr_of_lane):
    worst = 1
    for g0 in range(0, 64, 8):
        slots = {}
        for l in range(g0, g0 + 8):
            a = addr_of_lane(l)
            slots.setdefault((a // 16) % 8, set()).add(a)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def win_off(px, g):
    return px * 128 + ((g ^ (((px & 1) << 2) | ((px >> 1) & 3))) * 16)


worst_ww = max(write_conflicts(lambda l: win_off(l & 31, 2 * q + (l >> 5))) for q in range(4))
worst_wr = max(conflicts(lambda l: win_off(t * 16 + (l >> 2), 2 * (l & 3) + h)) for t in range(2) for h in range(2))
worst_wp = max(conflicts(lambda l: win_off(2 * (l >> 2) + (d ^ ((l >> 4) & 1)), 2 * (l & 3) + h)) for d in range(2) for h in range(2))
print(f"epilogue window: writes {worst_ww}-way, read-back {worst_wr}-way, pooling reads {worst_wp}-way")
assert worst_ww == 1 and worst_wr == 1 and worst_wp == 1
