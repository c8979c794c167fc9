"""Next-round preparation (CPU only): how far would a closed-form 3x3 symmetric eigen-solver be from the cyclic Jacobi that oracle and device share
today?  The plane fit's Jacobi is ~11 k of its ~44 k cycles; the closed form (trigonometric eigenvalues, eigenvector of the smallest one
from cross products of the rows of A - lambda I) is ~40 dependent double operations + one acos/cos + one sqrt.  Truth = numpy eigh.
usage: python tools/eigen_study.py"""
import numpy as np

rng = np.random.default_rng(7)


def jacobi(A, sweeps=30):
    A = A.copy(); V = np.eye(3)
    for _ in range(sweeps):
        off = abs(A[0, 1]) + abs(A[0, 2]) + abs(A[1, 2])
        if off < 1e-300:
            break
        for p, q in ((0, 1), (0, 2), (1, 2)):
            if abs(A[p, q]) < 1e-300:
                continue
            theta = (A[q, q] - A[p, p]) / (2 * A[p, q])
            t = np.sign(theta) / (abs(theta) + np.sqrt(theta * theta + 1)) if theta != 0 else 1.0
            c = 1 / np.sqrt(t * t + 1); s = t * c
            J = np.eye(3); J[p, p] = c; J[q, q] = c; J[p, q] = s; J[q, p] = -s
            A = J.T @ A @ J; V = V @ J
    return np.diag(A).copy(), V


def closed_form(A):
    # eigenvalues: Smith 1961 / Kopp 2008; eigenvector of the smallest: the largest cross product of two rows of (A - l I)
    q = np.trace(A) / 3
    B = A - q * np.eye(3)
    p = np.sqrt(np.sum(B * B) / 6)
    if p < 1e-300:
        return np.array([q, q, q]), np.array([0.0, 0.0, 1.0])
    r = np.clip(np.linalg.det(B / p) / 2, -1, 1)
    phi = np.arccos(r) / 3
    l_max = q + 2 * p * np.cos(phi)
    l_min = q + 2 * p * np.cos(phi + 2 * np.pi / 3)
    l_mid = 3 * q - l_max - l_min
    M = A - l_min * np.eye(3)
    cands = [np.cross(M[0], M[1]), np.cross(M[0], M[2]), np.cross(M[1], M[2])]
    n = max(cands, key=lambda v: v @ v)
    return np.array([l_min, l_mid, l_max]), n / np.linalg.norm(n)


def cluster(kind):
    n = int(rng.integers(6, 60))
    if kind == "plane":      # a wall / ground patch inside a 0.5 m voxel, range noise 2 cm
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        pts = np.c_[rng.uniform(-0.25, 0.25, n), rng.uniform(-0.25, 0.25, n), rng.normal(0, 0.02, n)] @ R.T
    elif kind == "edge":     # two planes meeting: lambda_min near the planarity threshold of 0.01
        a = np.c_[rng.uniform(-0.25, 0.25, n), rng.uniform(0, 0.25, n), rng.normal(0, 0.02, n)]
        b = np.c_[rng.uniform(-0.25, 0.25, n), rng.normal(0, 0.02, n), rng.uniform(0, 0.35, n)]
        pts = np.r_[a, b]
    else:                    # clutter
        pts = rng.uniform(-0.25, 0.25, (n, 3))
    pts = pts + rng.uniform(-300, 300, 3)      # world coordinates: the moment sums lose digits exactly as in init_plane
    c = pts.mean(0)
    return (pts.T @ pts) / len(pts) - np.outer(c, c)


def main():
    worst = {"jacobi": [0, 0], "closed": [0, 0]}
    flips = {"jacobi": 0, "closed": 0}
    N = 20000
    for i in range(N):
        A = cluster(("plane", "edge", "clutter")[i % 3])
        w, V = np.linalg.eigh(A)
        for name, (ev, nvec) in (("jacobi", (lambda r: (np.sort(r[0]), r[1][:, np.argmin(r[0])]))(jacobi(A))), ("closed", closed_form(A))):
            lmin = np.min(ev)
            worst[name][0] = max(worst[name][0], abs(lmin - w[0]) / max(w[2], 1e-300))
            if w[1] - w[0] > 1e-3 * w[2]:       # the normal is only defined when the smallest eigenvalue is separated
                worst[name][1] = max(worst[name][1], 1 - abs(nvec @ V[:, 0]))
            flips[name] += int((lmin < 0.01) != (w[0] < 0.01))
    for name in worst:
        print(f"{name:7s} max |lambda_min error| / lambda_max = {worst[name][0]:.2e}   max (1 - |n . n_true|) = {worst[name][1]:.2e}   "
              f"planarity decisions (lambda_min < 0.01) differing from eigh: {flips[name]} of {N}")


if __name__ == "__main__":
    main()
