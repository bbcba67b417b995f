"""CPU tier: the deal.II ParameterHandler reader (adaflo_amd.flow_parameters_from_prm) against
FlowParameters::parse_parameters (source/parameters.cc:449-614).  The input below is written in
the format of the reference's tests/*.prm with the values of its 2D rising-bubble case."""
import pytest

import adaflo_amd
from adaflo_amd.parameters import parse_prm

RISING_BUBBLE = """
# Listing of Parameters
subsection Two phase
  set density              = 1.
  set density difference   = -0.9
  set viscosity            = 0.01
  set viscosity difference = -0.009
  set surface tension      = 0.0245
  set epsilon              = 1.5
  set gravity              = 0.98
  set concentration subdivisions = 4
  set grad pressure compatible = 1
  set curvature correction = 1
  set number reinit steps = 2
  set number initial reinit steps = 2
end
subsection Time stepping
  set scheme           = bdf_2
  set end time         = 3
  set step size        = 0.02
end
subsection Navier-Stokes
  set dimension            = 2
  set global refinements   = 3   # 40 x 80 cells
  set adaptive refinements = 0
  set velocity degree      = 2
  subsection Solver
    set linearization scheme         = coupled implicit Picard
    set NL max iterations            = 10
    set NL tolerance                 = 1.e-9
    set lin max iterations           = 30
    set lin tolerance                = 1.e-4
    set lin velocity preconditioner  = ilu
    set lin its before inner solvers = 50
  end
end
subsection Output options
  set output filename  = output-rising_bubble_ls/data
  set output verbosity = 1
end
"""


def test_rising_bubble_parameter_file():
    p = adaflo_amd.flow_parameters_from_prm(RISING_BUBBLE)
    assert (p.dimension, p.velocity_degree, p.global_refinements) == (2, 2, 3)
    assert (p.density, p.density_diff, p.viscosity, p.viscosity_diff) == (1.0, -0.9, 0.01, -0.009)
    assert (p.surface_tension, p.epsilon, p.gravity) == (0.0245, 1.5, 0.98)
    assert p.concentration_subdivisions == 4 and p.interpolate_grad_onto_pressure and p.curvature_correction
    assert (p.n_reinit_steps, p.n_initial_reinit_steps) == (2, 2)
    assert (p.time_step_scheme, p.end_time, p.time_step_size_start) == ("bdf_2", 3.0, 0.02)
    assert p.linearization == "coupled implicit Picard"
    assert (p.max_nl_iteration, p.tol_nl_iteration, p.max_lin_iteration, p.tol_lin_iteration) == (10, 1e-9, 30, 1e-4)
    assert p.rel_lin_iteration                                   # default of the reference: 1
    # accepted but not used by the engine
    assert p.unused[("Navier-Stokes/Solver", "lin velocity preconditioner")] == "ilu"
    assert ("Output options", "output verbosity") in p.unused
    ts = adaflo_amd.TimeStepping(p)
    ts.next()
    assert abs(ts.weight() - 50.0) < 1e-12                       # first BDF-2 step = implicit Euler, 1 / dt


def test_defaults_and_reference_rules():
    p = adaflo_amd.flow_parameters_from_prm("subsection Navier-Stokes\n  set physical type = stokes\nend\n")
    assert p.density == 0.0                                      # parameters.cc:477-478
    assert (p.max_nl_iteration, p.tol_nl_iteration, p.max_lin_iteration, p.tol_lin_iteration) == (10, 1e-6, 500, 1e-3)
    # the Two phase section overrides density / viscosity only when positive (:548-557)
    p = adaflo_amd.flow_parameters_from_prm(
        "subsection Navier-Stokes\n set viscosity = 0.3\nend\nsubsection Two phase\n set viscosity = 0\nend\n")
    assert p.viscosity == 0.3
    # min step size above the start step size disables the adaptive step size (:593-595)
    p = adaflo_amd.flow_parameters_from_prm(
        "subsection Time stepping\n set step size = 0.1\n set min step size = 0.5\n set max step size = 2\nend\n")
    assert p.time_step_size_min == p.time_step_size_max == 0.1


def test_errors():
    with pytest.raises(ValueError):
        parse_prm("subsection A\n set x = 1\n")                 # unclosed
    with pytest.raises(ValueError):
        parse_prm("end\n")
    with pytest.raises(ValueError):
        parse_prm("subsection A\n bogus line\nend\n")
    with pytest.raises(ValueError):                              # "Linearization ... not available"
        adaflo_amd.flow_parameters_from_prm(
            "subsection Navier-Stokes\n subsection Solver\n  set linearization scheme = something\n end\nend\n")
    with pytest.raises(NotImplementedError):                     # velocity degree > 1
        adaflo_amd.flow_parameters_from_prm("subsection Navier-Stokes\n set velocity degree = 1\nend\n")


def test_time_stepping_helpers():
    """TimeStepping::set_desired_time_step / restart / at_tick / name (source/time_stepping.cc:104-119,206-268)"""
    p = adaflo_amd.FlowParameters(time_step_size_start=0.1, time_step_size_min=0.02, time_step_size_max=0.3, end_time=2.0)
    ts = adaflo_amd.TimeStepping(p)
    assert ts.name() == "BDF-2"
    ts.set_desired_time_step(1.0)                # at t = 0 the desired value counts as the previous one: only the bounds act
    assert ts.step_size() == 0.3
    ts.next()
    ts.set_desired_time_step(1.0)                # now limited to twice the previous step ... and to the maximum
    assert ts.step_size() == 0.3
    ts.set_desired_time_step(0.01)               # not below half the previous step
    assert abs(ts.step_size() - 0.15) < 1e-15
    ts.set_desired_time_step(0.001)
    ts.set_desired_time_step(0.001)
    ts.set_desired_time_step(0.001)
    assert ts.step_size() == 0.02                # ... and not below the minimum
    ts.restart()
    assert (ts.now(), ts.step_no(), ts.step_size(), ts.old_step_size(), ts.at_end()) == (0.0, 0, 0.1, 0.0, False)
    q = adaflo_amd.FlowParameters(time_step_size_start=0.05, end_time=1.0)
    t2 = adaflo_amd.TimeStepping(q)
    ticks = []
    for _ in range(8):
        t2.next()
        ticks.append(t2.at_tick(0.2))
    assert ticks == [False, False, False, True, False, False, False, True]     # output every 0.2: steps 4, 8 (beltrami_3d.prm)
