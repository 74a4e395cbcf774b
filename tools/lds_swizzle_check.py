import itertools
# LDS image: [row][64 B] (4 slots of 16 B). physical address(row, slot) = prow(row)*64 + pslot(row, slot)*16
# READ (ds_read_b128): lane l reads row r0 + (l&15), slot (l>>4); lane groups per the guide; bank set = 256 B line (16 slots of 16 B)
RGROUPS = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
           list(range(4,12))+list(range(16,20))+list(range(28,32)),
           list(range(32,36))+list(range(44,48))+list(range(52,60)),
           list(range(36,44))+list(range(48,52))+list(range(60,64))]
def read_conf(prow, pslot, r0):
    worst = 1
    for g in RGROUPS:
        cnt = {}
        for l in g:
            row, slot = r0 + (l & 15), l >> 4
            a = prow(row)*64 + pslot(row, slot)*16
            b = (a // 16) % 16
            cnt[b] = cnt.get(b, 0) + 1
        worst = max(worst, max(cnt.values()))
    return worst
# WRITE (ds_write_b128): contiguous 8-lane groups, bank = 128 B window (8 slots of 16 B).
# wgrad staging: unit u = tid..: lane i in group -> cg = cg0 + i, same ks; at instruction c writes row cg*8+c slot ks
def write_conf_wgrad(prow, pslot):
    worst = 1
    for cg0 in range(0, 16, 8):
        for c in range(8):
            for ks in range(4):
                cnt = {}
                for i in range(8):
                    row = (cg0 + i)*8 + c
                    a = prow(row)*64 + pslot(row, ks)*16
                    b = (a // 16) % 8
                    cnt[b] = cnt.get(b, 0) + 1
                worst = max(worst, max(cnt.values()))
    return worst
# conv staging: lane = (row = t>>2, seg = t&3): group of 8 lanes = 2 consecutive rows x 4 slots
def write_conf_conv(prow, pslot):
    worst = 1
    for r in range(0, 64, 2):
        cnt = {}
        for i in range(8):
            row, slot = r + (i >> 2), i & 3
            a = prow(row)*64 + pslot(row, slot)*16
            b = (a // 16) % 8
            cnt[b] = cnt.get(b, 0) + 1
        worst = max(worst, max(cnt.values()))
    return worst
cands = {}
cands['current'] = (lambda r: r, lambda r, s: s ^ ((-(r >> 2)) & 3))
cands['A'] = (lambda r: r ^ ((r >> 3) & 1), lambda r, s: s ^ ((-(r >> 2)) & 3) ^ ((r >> 4) & 3))
cands['B'] = (lambda r: r, lambda r, s: s ^ ((-(r >> 2)) & 3) ^ ((r >> 4) & 3))
cands['C'] = (lambda r: r ^ ((r >> 3) & 1), lambda r, s: s ^ ((-(r >> 2)) & 3) ^ ((r >> 4) & 3) ^ 0)
cands['D'] = (lambda r: r ^ ((r >> 3) & 1), lambda r, s: s ^ (((-(r >> 2)) ^ (r >> 4) ^ (r>>6)) & 3))
for k, (pr, ps) in cands.items():
    rc = max(read_conf(pr, ps, r0) for r0 in range(0, 128, 16))
    print(k, "read", rc, "write_wgrad", write_conf_wgrad(pr, ps), "write_conv", write_conf_conv(pr, ps))
