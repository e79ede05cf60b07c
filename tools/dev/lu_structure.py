"""Developer aid (CPU): structure of the symbolic LDU as the engine orders it -- how the LDS pivots of the trailing columns are spread."""
import os, sys, importlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
R = importlib.import_module("rac-2d_amd")
net = R.Network(os.path.join(ROOT, "data", "rate06_dipole_reformated_again_withgrain_lowH2Bind.dat"))
n = net.nSpecies
cp, ri = net.jac_pattern()
perm, fd = net.lu_ordering()
ns = fd - 1
inv = np.zeros(n, int); inv[perm - 1] = np.arange(n)
A = np.zeros((n, n), bool)
for j in range(n):
    for q in range(cp[j] - (1 if cp[0] == 1 else 0), cp[j + 1] - (1 if cp[0] == 1 else 0)):
        i = ri[q] - (1 if ri.min() >= 1 else 0)
        A[inv[i], inv[j]] = True
A |= np.eye(n, dtype=bool)
F = A.copy()
F[ns:, ns:] = True
for k in range(n):
    r = np.nonzero(F[k + 1:, k])[0] + k + 1
    c = np.nonzero(F[k, k + 1:])[0] + k + 1
    F[np.ix_(r, c)] = True
L = np.tril(F, -1); U = np.triu(F, 1)
print("n", n, "ns", ns, "L11", L[:ns, :ns].sum(), "L21", L[ns:, :ns].sum(), "U11", U[:ns, :ns].sum(), "U12", U[:ns, ns:].sum())
U12 = U[:ns, ns:]; L21 = L[ns:, :ns]; L11 = L[:ns, :ns]
rows_used = np.nonzero(U12.any(1))[0]
print("rows k of U12 with a nonzero:", len(rows_used), "of", ns, "; their mean density over the 121 columns: %.2f" % U12[rows_used].mean())
lenL = L.sum(0)
print("madds of the LDS pivots of trailing columns:", int((U12 * lenL[:ns, None]).sum()), " of all columns k<ns:", int((U[:ns, :ns] * lenL[:ns, None]).sum()))
print("L loads if every L column k were read once per group of G trailing columns:")
for G in (1, 2, 3, 4, 6, 8, 12):
    tot = 0
    for g0 in range(0, n - ns, G):
        anyk = U12[:, g0:g0 + G].any(1)
        tot += int(lenL[:ns][anyk].sum())
    print("  G=%2d: %7d entries (%.0f KB)" % (G, tot, tot * 8 / 1024))
dens = U12[rows_used].sum(1)
print("histogram of nonzeros per used U12 row:", np.histogram(dens, bins=[1, 2, 5, 10, 30, 60, 100, 122])[0])
lenL21 = L21.sum(0); lenL11 = L11.sum(0)
print("pivot-weighted mean len: L11 part %.1f, L21 part %.1f" % ((U12 * lenL11[:, None]).sum() / U12.sum(), (U12 * lenL21[:, None]).sum() / U12.sum()))
print("---- panel scheme (G = 12 trailing columns per panel)")
G = 12
n11 = L11.sum(0); n21 = L21.sum(0)
pairs11 = int((U12 & (n11[:, None] > 0)).sum())
print("phase 1: (k, j) pairs with a non-empty L11 piece:", pairs11, "of", int(U12.sum()), "; L11 entries read:", int((U12 * n11[:, None]).sum()))
vis = 0; ent = 0; mx = 0; pairs2 = 0
for g0 in range(0, n - ns, G):
    blk = U12[:, g0:g0 + G]
    anyk = blk.any(1) & (n21 > 0)
    vis += int(anyk.sum()); ent += int(n21[anyk].sum()); mx = max(mx, int(blk.sum())); pairs2 += int(blk[anyk].sum())
print("phase 2: k-visits", vis, " L21 entries read", ent, " pairs", pairs2, " max U12 entries of a panel", mx)
lenA = L[ns:ns + 64, :ns].sum(0); lenB = L[ns + 64:, :ns].sum(0)
print("L21 pieces: with A rows %d, with B rows %d, max len A %d B %d" % ((lenA > 0).sum(), (lenB > 0).sum(), lenA.max(), lenB.max()))
print("---- entry-parallel level ops: per column j, its pivots k < ns by level within the column; one op = up to 64 (pivot, L entry) pairs of one level")
ops = 0; ents = 0; piv = 0; lev_tot = 0
hist = []
for j in range(n):
    rows_u = np.nonzero(U[:min(j, ns), j])[0] if j > 0 else np.array([], int)
    if len(rows_u) == 0:
        continue
    plev = {}
    for a in rows_u:
        l = 0
        for b in rows_u:
            if b >= a: break
            if F[a, b]: l = max(l, plev[b] + 1)
        plev[a] = l
    nl = max(plev.values()) + 1
    lev_tot += nl
    for l in range(nl):
        ks = [k for k in rows_u if plev[k] == l]
        E = int(sum(lenL[k] for k in ks))
        piv += len(ks); ents += E
        o = -(-E // 64) if E > 0 else 0
        ops += o; hist.append(E)
print("columns with pivots: levels %d, pivots %d, entries %d -> ops %d (now: one op per pivot = %d); mean fill of an op %.1f of 64" % (lev_tot, piv, ents, ops, piv, ents / max(ops, 1)))
