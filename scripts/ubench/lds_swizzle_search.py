"""Exhaustive search behind r8v_px (csrc/res8v_kernels.h): which XOR-linear permutation g(P) of the four 16-byte pieces of a
64-byte pixel-pair record makes the window reads of the vector-ALU level-0 kernels conflict-free on gfx950?

A lane p (0..31 per row) reads the same piece of pair P = p + b (b = window offset); ds_read_b128 is served in the lane groups
G1 = {0-3, 12-15, 20-27}, G2 = {4-11, 16-19, 28-31} (MI355X_MICROARCH.md, LDS table), 64 banks x 4 B: 16 lanes must hit 16 different
16-byte bank quads, quad = (4 P + (k ^ g(P))) mod 16.  ds_write_b128 is served in runs of 8 consecutive lanes over 32 banks.
Prints the conflict counts (extra cycles summed over the 8 offsets and both groups) of r8_px's term and of the best candidates."""
G1 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
G2 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]


def read_conflicts(g):
    tot = 0
    for b in range(8):
        for grp in (G1, G2):
            seen = {}
            for lane in grp:
                P = lane + b
                q = (4 * (P & 3) + g(P)) % 16
                seen[q] = seen.get(q, 0) + 1
            tot += sum(v - 1 for v in seen.values())
    return tot


def write_conflicts(g):
    tot = 0
    for b in range(8):
        for m in range(4):
            seen = {}
            for lane in range(8 * m, 8 * m + 8):
                P = lane + b
                u = (4 * (P & 1) + g(P)) % 8
                seen[u] = seen.get(u, 0) + 1
            tot += sum(v - 1 for v in seen.values())
    return tot


def parity(v):
    return bin(v).count("1") & 1


if __name__ == "__main__":
    old = lambda P: (P >> 1) & 3
    print("r8_px  g = (P >> 1) & 3:            reads", read_conflicts(old), "writes", write_conflicts(old))
    new = lambda P: ((P >> 2) & 1) | ((((P >> 1) ^ (P >> 3)) & 1) << 1)
    print("r8v_px g = b2 | (b1 ^ b3) << 1:     reads", read_conflicts(new), "writes", write_conflicts(new))
    res = []
    for m0 in range(1, 64):
        for m1 in range(m0 + 1, 64):
            g = lambda P, m0=m0, m1=m1: parity(P & m0) | (parity(P & m1) << 1)
            r = read_conflicts(g)
            if r == 0:
                res.append((write_conflicts(g), m0, m1))
    res.sort()
    print(len(res), "read-conflict-free XOR-linear candidates; fewest write conflicts:", res[:6])
