"""numpy restatement of the Krylov recurrences the engine runs on the device (csrc/krylov.hip).

TEST INFRASTRUCTURE ONLY (like everything under oracle/).  deal.II's SolverCG / SolverBicgstab /
ReductionControl are not part of the reference tree (unpinned third-party dependency, >= 9.3);
these are the published algorithms they implement: Hestenes-Stiefel preconditioned CG and
van der Vorst's preconditioned BiCGStab with a convergence check after the first half step.
A is a callable x -> A x, Pinv a vector (DiagonalPreconditioner) or None."""
import numpy as np


def _converged(res, res0, abs_tol, rel_tol):
    return res <= abs_tol or res <= rel_tol * res0


def cg(A, b, x0=None, inv_diag=None, max_it=2000, abs_tol=1e-50, rel_tol=1e-6):
    x = np.zeros_like(b) if x0 is None else x0.copy()
    P = (lambda v: v * inv_diag) if inv_diag is not None else (lambda v: v.copy())
    r = b - A(x)
    res0 = res = np.linalg.norm(r)
    if _converged(res, res0, abs_tol, rel_tol):
        return x, 0, res0, res, True
    z = P(r)
    p = z.copy()
    rz = r @ z
    for it in range(1, max_it + 1):
        Ap = A(p)
        alpha = rz / (p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        res = np.linalg.norm(r)
        if _converged(res, res0, abs_tol, rel_tol):
            return x, it, res0, res, True
        z = P(r)
        rz_new = r @ z
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, max_it, res0, res, False


def bicgstab(A, b, x0=None, inv_diag=None, max_it=30, abs_tol=1e-50, rel_tol=1e-8):
    x = np.zeros_like(b) if x0 is None else x0.copy()
    P = (lambda v: v * inv_diag) if inv_diag is not None else (lambda v: v.copy())
    r = b - A(x)
    rbar = r.copy()
    res0 = res = np.linalg.norm(r)
    if _converged(res, res0, abs_tol, rel_tol):
        return x, 0, res0, res, True
    rho = alpha = omega = 1.0
    p = v = None
    it = 0
    for it in range(1, max_it + 1):
        rho_new = rbar @ r
        if rho_new == 0.0 or omega == 0.0:
            return x, it, res0, res, False
        if it == 1:
            p = r.copy()
        else:
            beta = (rho_new / rho) * (alpha / omega)
            p = r + beta * (p - omega * v)
        rho = rho_new
        y = P(p)
        v = A(y)
        alpha = rho / (rbar @ v)
        r -= alpha * v
        res = np.linalg.norm(r)
        if _converged(res, res0, abs_tol, rel_tol):
            x += alpha * y
            return x, it, res0, res, True
        z = P(r)
        t = A(z)
        omega = (t @ r) / (t @ t)
        x += alpha * y + omega * z
        r -= omega * t
        res = np.linalg.norm(r)
        if _converged(res, res0, abs_tol, rel_tol):
            return x, it, res0, res, True
    return x, it, res0, res, False
