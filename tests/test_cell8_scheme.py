"""The lane algebra of csrc/cell8_dpp.h (not gpu): the DPP all-gather / reduce-scatter over the 8 lanes of a hexahedral cell,
emulated with NumPy permutations. Checks what the kernels rely on: register j of lane q holds the value of lane q ^ X(j), that
lane owns node (4j + t) ^ M(q) - t + ... i.e. node(j, t, q) = M(q ^ X(j)) + t, and after the three reduce rounds lane q holds the
sum over the cell's lanes of the partials for ITS nodes M(q) .. M(q) + 3; the padded table stride spreads the 8 rows a cell reads
in one instruction over 8 different 4-bank slots."""
import numpy as np

Q = np.arange(8)


def M(q):
    return ((q >> 2) & 1) * 28 ^ ((q >> 1) & 1) * 8 ^ (q & 1) * 4


def X(j):
    return (7 ^ (j & 3)) if j & 4 else j


# the three lane permutations as "lane q receives the value of lane perm[q]"
XOR1, XOR2, HALF_MIRROR = Q ^ 1, Q ^ 2, 7 - Q          # quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror


def all_gather(x):
    """c8_all_gather: x[q] -> u[j][q]"""
    u = [None] * 8
    u[0] = x
    u[1] = u[0][XOR1]
    u[2] = u[0][XOR2]
    u[3] = u[1][XOR2]
    for j in range(4):
        u[4 + j] = u[j][HALF_MIRROR]
    return u


def reduce_scatter(p):
    """c8_reduce_scatter: p[j][q] -> sum held by lane q"""
    p = [a.copy() for a in p]
    for j in range(4):
        p[j] = p[j] + p[4 + j][HALF_MIRROR]
    for j in range(2):
        p[j] = p[j] + p[2 + j][XOR2]
    return p[0] + p[1][XOR1]


def test_owner_map_is_a_bijection_onto_the_32_padded_nodes():
    owned = sorted(int(M(q)) + t for q in Q for t in range(4))
    assert owned == list(range(32))
    assert [int(M(q ^ 1) ^ M(q)) for q in Q] == [4] * 8 and [int(M(q ^ 2) ^ M(q)) for q in Q] == [8] * 8
    assert [int(M(q ^ 7) ^ M(q)) for q in Q] == [16] * 8


def test_all_gather_delivers_lane_q_xor_X_and_its_node():
    x = np.array([10.0 * q for q in Q])
    u = all_gather(x)
    for j in range(8):
        for q in Q:
            src = q ^ X(j)
            assert u[j][q] == x[src]
            for t in range(4):
                assert ((4 * j + t) ^ int(M(q))) == int(M(src)) + t      # the table row the kernel multiplies it with


def test_reduce_scatter_sums_every_node_over_the_cell_in_the_owner_lane():
    rng = np.random.default_rng(0)
    for t in range(4):
        contrib = rng.normal(size=(8, 32))                  # contrib[q][node]: lane q's partial for a node
        p = [np.array([contrib[q][(4 * j + t) ^ int(M(q))] for q in Q]) for j in range(8)]
        got = reduce_scatter(p)
        for q in Q:
            assert np.isclose(got[q], contrib[:, int(M(q)) + t].sum(), rtol=1e-15, atol=1e-15)


def test_padded_table_stride_is_bank_conflict_free_for_a_cell():
    qstride = 32 * 4 + 2                                    # C8_QSTRIDE, doubles
    for j in range(8):
        for t in range(4):
            rows = [(q * qstride + (((4 * j) ^ int(M(q))) + t) * 4) * 2 for q in Q]       # dword address of the row's first double
            slots = {(a // 4) % 16 for a in rows}            # 16-byte reads: 16 slots of 4 banks
            assert len(slots) == 8
    natural = [(q * 128 + ((0 ^ int(M(q)))) * 4) * 2 for q in Q]
    assert len({(a // 4) % 16 for a in natural}) == 2        # what the unpadded stride did: two bank positions, 4-way conflict
