#!/usr/bin/env python3
"""Blueprint of the diagonal-band fill proposed in DESIGN.md 8.8 (NOT a kernel; a schedule model in numpy float32).

The alpha / beta matrices of one (read, window) pair are filled on the band of diagonals d = j - i in [dlo, dhi] only (SPEC candidate `fill_band`, oracle knob
orc_set_fill_band; profiles/r04_fill_band_study.txt: bit-identical results).  On the GPU a LANE would own two adjacent diagonals, dlo + 2m and dlo + 2m + 1, and walk a
staircase: at anti-diagonal step s = i + j it holds the cell of the one of its two diagonals whose parity matches s, so every lane of a read has a cell at every step
(ceil(BW / 2) lanes per read, seven reads of 17 diagonals per wave64 sweep of I + J + 1 steps).  What a lane needs at step s:
    own value of step s - 1, own value of step s - 2 (the diagonal neighbour), ONE value of step s - 1 from a neighbouring lane (lane m - 1 when it is on its even
    diagonal, lane m + 1 when on its odd one): a single DPP row shift per step, alternating direction.
`fill_by_columns` is the oracle's order (oracle/ccs_oracle.c fill(), same operations, same rounding); `fill_by_staircase` is the lane / step schedule.  Each cell's
arithmetic is identical, so the two must agree bit for bit — tests/test_oracle_hmm.py::test_diagonal_band_schedule checks that, which pins the index arithmetic (lane <->
diagonal, step <-> cell, which neighbour comes from where, the boundary cells) a kernel has to get right.
"""
import numpy as np

F = np.float32


def band(I, J, fill_band, score_band=5):
    dIJ = abs(I - J)
    W = fill_band + score_band + max(0, dIJ - 2)
    return min(0, J - I) - W, max(0, J - I) + W


def fill_by_columns(ME, INS, DL, k, o, I, J, dlo, dhi):
    """column-major, as the oracle: gamma / alpha forward, beta backward; cells off the band are exact zeros"""
    gam = np.zeros((I + 2, J + 1), F); alp = np.zeros((I + 2, J + 1), F); bet = np.zeros((I + 2, J + 1), F)
    for j in range(J + 1):
        for i in range(I + 1):
            if not dlo <= j - i <= dhi:
                continue
            if j == 0:
                g = F(1.0) if i == 0 else F(0.0)
            else:
                m = F(alp[i - 1, j - 1] * ME[k[j - 1], o[i - 1]]) if i > 0 else F(0.0)
                g = F(m + F(alp[i, j - 1] * DL[k[j - 1]]))
            gam[i, j] = g
            st = F(alp[i - 1, j] * INS[k[j], o[i - 1]]) if (i > 0 and j < J) else F(0.0)
            alp[i, j] = F(g + st)
    bet[I, J] = F(1.0)
    for j in range(J - 1, -1, -1):
        for i in range(I, -1, -1):
            if not dlo <= j - i <= dhi:
                continue
            t1 = F(ME[k[j], o[i]] * bet[i + 1, j + 1]) if i < I else F(0.0)
            t2 = F(INS[k[j], o[i]] * bet[i + 1, j]) if i < I else F(0.0)
            t3 = F(DL[k[j]] * bet[i, j + 1])
            bet[i, j] = F(F(t1 + t2) + t3)
    return gam, alp, bet


def fill_by_staircase(ME, INS, DL, k, o, I, J, dlo, dhi):
    """the proposed lane / step schedule: lane m owns diagonals dlo + 2m (even phase) and dlo + 2m + 1 (odd phase)"""
    BW = dhi - dlo + 1
    nl = (BW + 1) // 2
    gam = np.zeros((I + 2, J + 1), F); alp = np.zeros((I + 2, J + 1), F); bet = np.zeros((I + 2, J + 1), F)
    z = F(0.0)
    # ---- alpha: steps s = 0 .. I + J; prev / prev2 = the lane's own values of the last two steps (0 where it had no cell)
    prev = [z] * nl; prev2 = [z] * nl
    for s in range(I + J + 1):
        p = (s - dlo) & 1
        cur = [z] * nl
        for m in range(nl):
            d = dlo + 2 * m + p
            if d > dhi or (s - d) % 2:
                continue                                   # (the lane's odd diagonal may lie outside an odd-width band)
            i, j = (s - d) // 2, (s + d) // 2
            if not (0 <= i <= I and 0 <= j <= J):
                continue
            nb = (prev[m - 1] if m > 0 else z) if p == 0 else (prev[m + 1] if m + 1 < nl else z)   # ONE shifted value per step
            left, up, diag = (nb, prev[m], prev2[m]) if p == 0 else (prev[m], nb, prev2[m])
            if j == 0:
                g = F(1.0) if i == 0 else z
            else:
                mm = F(diag * ME[k[j - 1], o[i - 1]]) if i > 0 else z
                g = F(mm + F(left * DL[k[j - 1]]))
            st = F(up * INS[k[j], o[i - 1]]) if (i > 0 and j < J) else z
            gam[i, j] = g
            cur[m] = F(g + st)
            alp[i, j] = cur[m]
        prev2, prev = prev, cur
    # ---- beta: the same staircase walked backwards (steps s = I + J .. 0): the neighbours are the cells of steps s + 1 / s + 2
    prev = [z] * nl; prev2 = [z] * nl
    for s in range(I + J, -1, -1):
        p = (s - dlo) & 1
        cur = [z] * nl
        for m in range(nl):
            d = dlo + 2 * m + p
            if d > dhi or (s - d) % 2:
                continue
            i, j = (s - d) // 2, (s + d) // 2
            if not (0 <= i <= I and 0 <= j <= J):
                continue
            if j == J:
                cur[m] = F(1.0) if i == I else z           # (only cell (I, J) of column J is ever non-zero)
                bet[i, j] = cur[m]
                continue
            # (i + 1, j): step s + 1, diagonal d - 1; (i, j + 1): step s + 1, diagonal d + 1; (i + 1, j + 1): step s + 2, diagonal d
            nb = (prev[m - 1] if m > 0 else z) if p == 0 else (prev[m + 1] if m + 1 < nl else z)
            down, right, diag = (nb, prev[m], prev2[m]) if p == 0 else (prev[m], nb, prev2[m])
            t1 = F(ME[k[j], o[i]] * diag) if i < I else z
            t2 = F(INS[k[j], o[i]] * down) if i < I else z
            t3 = F(DL[k[j]] * right)
            cur[m] = F(F(t1 + t2) + t3)
            bet[i, j] = cur[m]
        prev2, prev = prev, cur
    return gam, alp, bet


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for trial in range(20):
        J = int(rng.integers(8, 32)); I = max(1, J + int(rng.integers(-6, 7)))
        ME = rng.random((16, 12)).astype(F) * F(0.3); INS = rng.random((16, 12)).astype(F) * F(0.1); DL = rng.random(16).astype(F) * F(0.1)
        k = rng.integers(0, 16, J + 1); o = rng.integers(0, 12, I + 1)
        dlo, dhi = band(I, J, int(rng.integers(0, 4)))
        a = fill_by_columns(ME, INS, DL, k, o, I, J, dlo, dhi); b = fill_by_staircase(ME, INS, DL, k, o, I, J, dlo, dhi)
        assert all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(a, b)), (trial, I, J, dlo, dhi)
        print(f"trial {trial}: I {I} J {J} band [{dlo}, {dhi}] -> {(dhi - dlo + 2) // 2} lanes: bit-identical, alpha(I,J) {a[1][I, J]:.3e} beta(0,0) {a[2][0, 0]:.3e}")
