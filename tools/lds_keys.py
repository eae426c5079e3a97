import itertools
groups = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
          list(range(4,12))+list(range(16,20))+list(range(28,32)),
          list(range(32,36))+list(range(44,48))+list(range(52,60)),
          list(range(36,44))+list(range(48,52))+list(range(60,64))]
def rows_act(fr, j): return j*16 + fr
def rows_nat(fr, j): return 16*j + fr
def rows_pair(fr, j): return 32*(j>>1) + 8*(fr>>2) + 4*(j&1) + (fr&3)
def rows_w(fr, j): return 16*(fr>>2) + 4*j + (fr&3)
def ok(rowf, key):
    for j in range(4):
        for g in groups:
            slots = set()
            for l in g:
                fr, fg = l & 15, l >> 4
                r = rowf(fr, j)
                slot = (4*(r % 4) + (fg ^ key(r))) % 16
                if slot in slots: return False
                slots.add(slot)
    return True
for name, rowf in [("act", rows_act), ("pair", rows_pair), ("w", rows_w)]:
    found = []
    for p, q in itertools.permutations(range(6), 2):
        for T in itertools.product(range(4), repeat=4):
            key = lambda r, p=p, q=q, T=T: T[((r>>p)&1) | (((r>>q)&1)<<1)]
            if ok(rowf, key): found.append((p,q,T))
    print(name, len(found), found[:6])
