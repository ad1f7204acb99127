"""Bank-conflict check of the LDS images used by the GEMM and attention kernels, under the gfx950 LDS model of
/opt/skills/guides/MI355X_MICROARCH.md §LDS (lane groups and bank modulus per instruction).
Prints the worst-case cycles per wave-instruction vs the conflict-free figure.  Dev tool; no GPU needed."""
from collections import defaultdict

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def cycles(addrs, groups, nbytes, bank_mod):
    """addrs: byte address per lane (64).  Returns total LDS cycles = sum over groups of max distinct-address count per bank."""
    tot = 0
    for g in groups:
        banks = defaultdict(set)
        for l in g:
            a = addrs[l]
            for d in range(0, nbytes, 4):
                banks[((a + d) // 4) % bank_mod].add((a + d) // 4)
        tot += max(len(v) for v in banks.values())
    return tot


def read_b128(addrs):
    return cycles(addrs, B128_GROUPS, 16, 64), 4


def read_b64(addrs):  # also ds_read_b64_tr_b16 (first-order model)
    return cycles(addrs, [list(range(0, 32)), list(range(32, 64))], 8, 64), 2


def write_b128(addrs):
    return cycles(addrs, [list(range(g * 8, g * 8 + 8)) for g in range(8)], 16, 32), 8


# ---- GEMM images (gemm.hip) ----
def lds_row_off(r, c):
    return r * 128 + ((c ^ ((r >> 1) & 7)) << 4)


def lds_kmaj_off(k, q):
    return k * 256 + ((q ^ ((k & 3) | (((k >> 3) & 1) << 2))) << 5)


# ---- attention image (attention.hip) ----
def att_off(r, c):
    return r * 128 + ((c ^ ((((r >> 1) & 1) << 2) | ((r >> 2) & 3))) << 4)


def report(name, res):
    got, ideal = res
    print(f"{name:58s} {got:3d} cycles (conflict-free = {ideal}){'  <-- CONFLICTS' if got > ideal else ''}")


if __name__ == "__main__":
    worst = {}
    # GEMM row layout: staging writes (thread t: chunk t&7, row t>>3 (+32 i)) per wave w
    for w in range(4):
        report(f"gemm row-tile ds_write_b128 wave {w}", write_b128([lds_row_off(((w * 64 + l) >> 3), (w * 64 + l) & 7) for l in range(64)]))
    for base in (0, 16, 64):
        for ks in (0, 1):
            report(f"gemm row-tile frag ds_read_b128 base {base} ks {ks}", read_b128([lds_row_off(base + (l & 15), ks * 4 + (l >> 4)) for l in range(64)]))
    for w in range(4):
        t = [w * 64 + l for l in range(64)]
        report(f"gemm kmaj-tile ds_write_b128 wave {w}", write_b128([lds_kmaj_off(x >> 4, (x & 15) >> 1) + (((x & 15) & 1) << 4) for x in t]))
    for base in (0, 16, 112):
        for ks in (0, 1):
            for half in (0, 4):
                ad = [lds_kmaj_off(ks * 32 + (l >> 4) * 8 + ((l & 15) >> 2) + half, base >> 4) + ((l & 15) & 3) * 8 for l in range(64)]
                report(f"gemm kmaj-tile tr-read base {base} ks {ks} +{half}", read_b64(ad))
    # t256 kernel: 32-index fragments on the row layout and on the second contraction-major swizzle
    def kmaj2(k, q):
        return k * 256 + ((q ^ (((k & 3) << 1) | ((k >> 2) & 1))) << 5)
    for base in (0, 32, 96):
        for sstep in range(4):
            report(f"t256 row frag32 ds_read_b128 base {base} s {sstep}", read_b128([lds_row_off(base + (l & 31), sstep * 2 + (l >> 5)) for l in range(64)]))
            for plus in (0, 4):
                ad = []
                for l in range(64):
                    G, s16, hi = l >> 4, l & 15, l >> 5
                    ad.append(kmaj2(sstep * 16 + hi * 8 + (s16 >> 2) + plus, (base >> 4) + (G & 1)) + (s16 & 3) * 8)
                report(f"t256 kmaj2 tr-read base {base} s {sstep} +{plus}", read_b64(ad))
    for w in range(4):
        t = [w * 64 + l for l in range(64)]
        # LDS-DMA writes are lane-linear by construction; check the register-staged equivalent anyway
        report(f"t256 kmaj2 ds_write_b128-equivalent wave {w}", write_b128([kmaj2(x >> 4, (x & 15) >> 1) + (((x & 15) & 1) << 4) for x in t]))
    # attention
    for w in range(4):
        t = [w * 64 + l for l in range(64)]
        report(f"attn tile ds_write_b128 wave {w}", write_b128([att_off(x >> 3, x & 7) for x in t]))
    for rb in (0, 32):
        for ds in range(4):
            report(f"attn row frag ds_read_b128 rb {rb} ds {ds}", read_b128([att_off(rb + (l & 31), ds * 2 + (l >> 5)) for l in range(64)]))
    for rbase in (0, 16, 32, 48):
        for cb in (0, 1):
            for plus in (0, 8):
                ad = []
                for l in range(64):
                    G, s = l >> 4, l & 15
                    ad.append(att_off(rbase + 4 * (G >> 1) + (s >> 2) + plus, cb * 4 + (G & 1) * 2 + ((s & 3) >> 1)) + (s & 1) * 8)
                report(f"attn tr-read rbase {rbase} cb {cb} +{plus}", read_b64(ad))
