import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import adaflo_amd
from common import BETA, LIN, PHYS, rel_l2
from golden_util import FixedTimeStepping, load, prm_dict
for name in ["ns_3d_q2_3x3x3_stokes", "ns_3d_q2_4x4x4_newton"]:
    d = load(name); p = prm_dict(d); k = 2
    fp = adaflo_amd.FlowParameters(velocity_degree=k, physical_type=PHYS[p["physical_type"]], linearization=LIN[p["linearization"]],
        formulation_convective_term=BETA[p["beta"]], viscosity=p["viscosity"], density=p["density"],
        damping=-p["damping"], tau_grad_div=p["tau_grad_div"], density_diff=p["density_diff"])
    mesh = adaflo_amd.BrickMesh([int(n) for n in d["ncell"]], tuple(d["lower"]), tuple(d["upper"]))
    op = adaflo_amd.NavierStokesMatrix(fp, mesh, dirichlet_faces_u=range(6), constrained_faces_p=[0])
    op.initialize(FixedTimeStepping(p), False)
    for variant in (1, 0):
        op.set_kernel_variant(variant)
        src = op.block_vector(d["src_u"], d["src_p"])
        dp = op.initialize_p_vector(d["src_p"])
        op.pressure_poisson_vmult(dp, src.block(1))
        print(name, variant, "poisson", rel_l2(dp.numpy(), d["pressure_poisson"]), dp.numpy()[:6], d["pressure_poisson"][:6])
        op.pressure_mass_vmult(dp, src.block(1))
        print(name, variant, "mass", rel_l2(dp.numpy(), d["pressure_mass"]))
