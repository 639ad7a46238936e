"""CPU simulation: would a Winograd F(8x8,3x3) tile (100 multiplies per 64 outputs: 12 % fewer MFMA FLOPs than F(6x6)) keep
f32-grade results?  Cook-Toom matrices (wincnn construction, sympy rationals) for several point sets, one 256-channel
3x3 convolution in f32 against a float64 direct convolution.  Result (LAB_NOTES.md 3.1b): no -- 6.7e-5 at best per layer,
ten times F(6x6)'s 6.8e-6, 2.5e-4 .. 1.9e-3 for the other point sets."""
import numpy as np
from fractions import Fraction as Fr
def cook_toom(m, r, pts):
    # F(m, r): n = m + r - 1 points incl. infinity (last)
    n = m + r - 1
    assert len(pts) == n - 1
    # Using the standard construction: A^T (m x n), G (n x r), B^T (n x n) with Y = A^T [(G g) * (B^T d)]
    # Build via Lagrange: follow wincnn (Lavin) construction
    import sympy
    from sympy import Rational, Matrix, symbols, Poly
    a = [Rational(p.numerator, p.denominator) for p in pts]
    x = symbols('x')
    def At(a, m, n): return Matrix(m, n, lambda i, j: a[j]**i)
    def A_(a, m, n): return Matrix(m, n, lambda i, j: a[i]**j)
    def T(a, n): return Matrix(Matrix.eye(n).col_insert(n, Matrix(n, 1, lambda i, j: -a[i]**n)))
    def Lx(a, n):
        f = []
        for i in range(n):
            p = Poly(1, x)
            for j in range(n):
                if j != i: p = p * Poly(x - a[j], x)
            f.append(p)
        return f
    def Fd(a, n):
        return Matrix(n, 1, lambda i, j: sympy.prod([(a[i] - a[k]) for k in range(n) if k != i]))
    def Fdiag(a, n):
        f = Fd(a, n); return Matrix(n, n, lambda i, j: f[i, 0] if i == j else 0)
    def FdiagPlus1(a, n):
        f = Fdiag(a, n - 1); f = f.col_insert(n - 1, Matrix.zeros(n - 1, 1)); f = f.row_insert(n - 1, Matrix(1, n, lambda i, j: 1 if j == n - 1 else 0)); return f
    def L(a, n):
        lx = Lx(a, n); f = Fd(a, n)
        return Matrix(n, n, lambda i, j: lx[i].nth(j) / f[i]).T
    def Bt(a, n): return L(a, n) * T(a, n)
    def B(a, n): return Bt(a, n - 1).row_insert(n - 1, Matrix(1, n, lambda i, j: 1 if j == n - 1 else 0))
    nn = n
    AT = At(a, m, nn - 1).col_insert(nn - 1, Matrix(m, 1, lambda i, j: 1 if i == m - 1 else 0))
    G = (A_(a, nn - 1, r).row_insert(nn - 1, Matrix(1, r, lambda i, j: 1 if j == r - 1 else 0)))
    fd = FdiagPlus1(a, nn)
    G = (fd.inv() * G) if False else G
    # wincnn: AT = A(a,m,n).T ; G = (A(a,n,r).T * Fdiag^-1).T ; BT = Fdiag * B(a,n).T
    f = Fdiag(a, nn - 1)
    Gm = (A_(a, nn - 1, r).T * f.inv()).T
    Gm = Gm.row_insert(nn - 1, Matrix(1, r, lambda i, j: 1 if j == r - 1 else 0))
    BT = FdiagPlus1(a, nn) * B(a, nn).T
    return np.array(AT.tolist(), dtype=np.float64), np.array(Gm.tolist(), dtype=np.float64), np.array(BT.tolist(), dtype=np.float64)

def test(m, pts, C=256, trials=3):
    AT, G, BT = cook_toom(m, 3, pts)
    n = m + 2
    rng = np.random.default_rng(0)
    errs = []
    for _ in range(trials):
        d = rng.standard_normal((C, n, n)).astype(np.float32)  # activations ~ N(0,1) after relu-ish
        d = np.maximum(d, 0)
        g = (rng.uniform(-1, 1, (C, 3, 3)) * np.sqrt(6.0 / (C * 9))).astype(np.float32)
        # reference in f64
        ref = np.zeros((m, m))
        for i in range(m):
            for j in range(m):
                ref[i, j] = (d[:, i:i+3, j:j+3].astype(np.float64) * g.astype(np.float64)).sum()
        AT32, G32, BT32 = AT.astype(np.float32), G.astype(np.float32), BT.astype(np.float32)
        U = np.einsum('ij,cjk,lk->cil', G32, g, G32).astype(np.float32)
        V = np.einsum('ij,cjk,lk->cil', BT32, d, BT32).astype(np.float32)
        M = (U * V).astype(np.float32).sum(0, dtype=np.float32)
        Y = (AT32 @ M @ AT32.T).astype(np.float32)
        # direct f32
        dir32 = np.zeros((m, m), np.float32)
        for i in range(m):
            for j in range(m):
                dir32[i, j] = (d[:, i:i+3, j:j+3] * g).sum(dtype=np.float32)
        errs.append((np.abs(Y - ref).max() / np.abs(ref).max(), np.abs(dir32 - ref).max() / np.abs(ref).max(), np.abs(V).max(), np.abs(U).max() / np.abs(g).max()))
    e = np.array(errs)
    print(f"F({m}x{m}) pts={[str(p) for p in pts]}: wino err {e[:,0].max():.2e}  direct f32 err {e[:,1].max():.2e}  |V|max {e[:,2].max():.1f} |U|/|g| {e[:,3].max():.2f}")

test(4, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2)])
test(6, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2), Fr(1,2), Fr(-1,2)])
test(8, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2), Fr(1,2), Fr(-1,2), Fr(4), Fr(-4)])
test(8, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2), Fr(1,2), Fr(-1,2), Fr(1,4), Fr(-1,4)])
test(8, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2), Fr(1,2), Fr(-1,2), Fr(3,2), Fr(-3,2)])
test(8, [Fr(0), Fr(1), Fr(-1), Fr(2), Fr(-2), Fr(1,2), Fr(-1,2), Fr(3,4), Fr(-3,4)])
test(8, [Fr(0), Fr(1), Fr(-1), Fr(1,2), Fr(-1,2), Fr(3,2), Fr(-3,2), Fr(3,4), Fr(-3,4)])
