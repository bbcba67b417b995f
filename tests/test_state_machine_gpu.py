"""Randomised API sequences against a model of the operator's public state (Weak #10).

The engine keeps the quadrature-point linearisation state in up to ten device buffers (generic, streaming copies of two
kernel families, nodal copies for the recompute mode, each with a frozen twin) behind validity flags; the reference keeps two
arrays -- the current state and the one frozen by fix_linearization_point (include/adaflo/navier_stokes_matrix.h:162-178,
source/navier_stokes_matrix.cc:349-375, 1144-1152).  The model below IS the reference's view: the canonical current
state, the frozen state, the current and frozen coefficients, the scheme; every operator application of a random call
sequence is compared with the oracle applied to the model."""
import numpy as np
import pytest

from common import Case, rel_l2
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-12


class Model:
    def __init__(self, cases, k):
        self.cases, self.k = cases, k                 # {scheme: Case} on the same mesh
        self.scheme = 0
        self.lin = np.zeros(cases[0].n_cells * cases[0].nq * 12)
        self.coef = (None, None, None)
        self.defined = 12                             # entries per point the state defines (4 after a Picard-type residual)
        self.frozen = None                            # (lin, coefficients, defined)

    @property
    def case(self):
        return self.cases[self.scheme]

    def meaningful(self, lin):
        """the entries of a state the current scheme defines (Newton: all twelve; Picard-type: u and div u in slot 3)"""
        v = lin.reshape(-1, 12)
        return v if self.scheme == 0 else v[:, :4]


def _sequence(seed, k, ncell, n_calls):
    rng = np.random.default_rng(seed)
    cases = {s: Case(ncell, k=k, linearization=s, tau_grad_div=0.1, damping=0.1, steps=3, seed=seed) for s in (0, 1)}
    m = Model(cases, k)
    c0 = cases[0]
    op = c0.engine()
    w, modes = c0.weights_modes()
    log = []
    vec_u = lambda: rng.uniform(-1, 1, c0.n_u)       # noqa: E731
    vec_p = lambda: rng.uniform(-1, 1, c0.n_p)       # noqa: E731
    # a defined start: a state through the front door
    m.lin = rng.uniform(-1, 1, m.lin.size)
    op.set_linearization(m.lin)
    log.append("set_linearization")
    calls = ["set_linearization", "residual", "set_coefficients", "clear_coefficients", "scheme", "variant", "fix", "vmult",
             "velocity_vmult", "get_linearization", "pressure_ops", "vmult", "velocity_vmult", "residual"]
    for _ in range(n_calls):
        call = calls[rng.integers(len(calls))]
        log.append(call)
        ctx = "seed %d k %d %s: %s" % (seed, k, ncell, " > ".join(log))
        co = dict(rho=m.coef[0], mu=m.coef[1], damp=m.coef[2])
        if call == "set_linearization":
            m.lin, m.defined = rng.uniform(-1, 1, m.lin.size), 12
            op.set_linearization(m.lin)
        elif call == "residual":
            su, sp, ou, oou = vec_u(), vec_p(), vec_u(), vec_u()
            lin = np.zeros(m.lin.size)
            ref = orc.ns_residual(m.case.mesh, k, m.case.prm, su, sp, ou, oou, con_u=m.case.con_u, con_p=m.case.con_p, lin=lin, **co)
            keep = m.lin.reshape(-1, 12).copy()
            keep[:, :(12 if m.scheme == 0 else 4)] = lin.reshape(-1, 12)[:, :(12 if m.scheme == 0 else 4)]
            m.lin = keep.reshape(-1)               # (a Picard-type residual writes u and div u only; the reference keeps what
            m.defined = 12 if m.scheme == 0 else 4 # the array held in the other entries -- the engine does not promise that)
            rhs = op.block_vector()
            op.residual(rhs, op.block_vector(su, sp), None, op.block_vector(ou), op.block_vector(oou))
            gu, gp = rhs.numpy()
            assert rel_l2(gu, ref[0]) < TOL and rel_l2(gp, ref[1]) < TOL, (ctx, rel_l2(gu, ref[0]), rel_l2(gp, ref[1]))
        elif call == "set_coefficients":
            m.coef = c0.random_coefficients()
            op.set_coefficients(*m.coef)
        elif call == "clear_coefficients":
            m.coef = (None, None, None)
            op.set_coefficients(None, None, None)
        elif call == "scheme":
            m.scheme = 1 - m.scheme
            op.parameters = m.case.fp
            op.update_parameters()
            if m.scheme == 0 and m.defined < 12:   # Newton cannot read a state that holds (u, div u) only: a new state
                m.lin, m.defined = rng.uniform(-1, 1, m.lin.size), 12
                op.set_linearization(m.lin)
                log.append("set_linearization")
        elif call == "variant":
            op.set_kernel_variant([0, 1, 4][rng.integers(3)])
        elif call == "fix":
            m.frozen = (m.lin.copy(), m.coef, m.defined)
            op.fix_linearization_point()
        elif call == "vmult":
            su, sp = vec_u(), vec_p()
            ref = orc.ns_vmult(m.case.mesh, k, m.case.prm, su, sp, m.case.con_u, m.case.con_p, lin=m.lin, weights=w, modes=modes, **co)
            dst = op.block_vector(np.full(c0.n_u, 7.0), np.full(c0.n_p, 7.0))
            op.vmult(dst, op.block_vector(su, sp))
            gu, gp = dst.numpy()
            assert rel_l2(gu, ref[0]) < TOL and rel_l2(gp, ref[1]) < TOL, (ctx, rel_l2(gu, ref[0]), rel_l2(gp, ref[1]))
        elif call == "velocity_vmult":
            su = vec_u()
            lin, cf, defined = m.frozen if m.frozen is not None else (m.lin, m.coef, m.defined)
            if m.scheme == 0 and defined < 12:
                continue                           # (frozen by a Picard-type residual, read by Newton: undefined)
            ref = orc.ns_velocity_vmult(m.case.mesh, k, m.case.prm, su, m.case.con_u, lin=lin, rho=cf[0], mu=cf[1], damp=cf[2])
            dst = op.initialize_u_vector(np.full(c0.n_u, 3.0))
            op.velocity_vmult(dst, op.initialize_u_vector(su))
            assert rel_l2(dst.numpy(), ref) < TOL, (ctx, rel_l2(dst.numpy(), ref))
        elif call == "get_linearization":
            got = op.get_linearization()
            assert rel_l2(m.meaningful(got), m.meaningful(m.lin)) < TOL, (ctx, rel_l2(m.meaningful(got), m.meaningful(m.lin)))
        elif call == "pressure_ops":
            sp = vec_p()
            # (both work on the FROZEN densities / viscosities when there are any, navier_stokes_matrix.cc:393-411, 429-441: "the
            # multiplication is done on the matrix the preconditioner is based upon")
            rho_p = m.frozen[1][0] if (m.frozen is not None and m.frozen[1][0] is not None) else m.coef[0]
            mu_p = m.frozen[1][1] if (m.frozen is not None and m.frozen[1][1] is not None) else m.coef[1]
            for name, ref in (("pressure_poisson_vmult", orc.ns_pressure_poisson_vmult(m.case.mesh, k, m.case.prm, sp, m.case.con_p, rho=rho_p)),
                              ("pressure_mass_vmult", orc.ns_pressure_mass_vmult(m.case.mesh, k, m.case.prm, sp, m.case.con_p, mu=mu_p))):
                dst = op.initialize_p_vector(np.full(c0.n_p, 5.0))
                getattr(op, name)(dst, op.initialize_p_vector(sp))
                assert rel_l2(dst.numpy(), ref) < TOL, (ctx, name, rel_l2(dst.numpy(), ref))
    op.clear()


@pytest.mark.parametrize("k,ncell,seeds", [(2, (5, 4, 3), range(0, 60)), (2, (9, 8, 5), range(100, 140)),
                                          (4, (5, 4, 3), range(200, 250)), (4, (9, 8, 5), range(300, 330)),
                                          (3, (4, 3, 3), range(400, 420))])
def test_random_api_sequences_against_the_reference_state_model(k, ncell, seeds):
    """200 seeded sequences of up to twelve calls: set_linearization, residual, set_coefficients (and clearing them), a change
    of scheme (Newton <-> Picard-type), set_kernel_variant (0, 1, 4), fix_linearization_point, vmult, velocity_vmult,
    get_linearization, the pressure sub-blocks"""
    for seed in seeds:
        _sequence(seed, k, ncell, 12)
