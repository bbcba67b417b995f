"""The subset of adaflo's FlowParameters the operator kernels read
(include/adaflo/parameters.h:55-90, source/parameters.cc:451-520)."""
from dataclasses import dataclass

PHYSICAL_TYPES = {"incompressible": 0, "incompressible stationary": 1, "stokes": 2}
LINEARIZATIONS = {"coupled implicit Newton": 0, "coupled implicit Picard": 1,
                  "coupled velocity semi-implicit": 2, "coupled velocity explicit": 3,
                  "projection": 4}
# get_beta_formulation_convective_term_momentum_balance, parameters.h:63-68
BETA = {"conservative": 1.0, "convective": 0.0, "skew-symmetric": 0.5}
TIME_SCHEMES = ("explicit_euler", "implicit_euler", "crank_nicolson", "bdf_2")


@dataclass
class FlowParameters:
    dimension: int = 3
    velocity_degree: int = 2
    physical_type: str = "incompressible"
    linearization: str = "coupled implicit Newton"
    formulation_convective_term: str = "skew-symmetric"
    viscosity: float = 1.0
    density: float = 1.0
    damping: float = 0.0          # as written in the .prm; the stored sign is flipped
    tau_grad_div: float = 0.0
    density_diff: float = 0.0
    viscosity_diff: float = 0.0
    augmented_taylor_hood: bool = False
    # time stepping section
    time_step_scheme: str = "bdf_2"
    start_time: float = 0.0
    end_time: float = 1.0
    time_step_size_start: float = 1e-2     # parameters.cc:377-410 (declare_entry defaults)
    time_step_size_max: float = 1.0
    time_step_size_min: float = 0.1
    # two-phase section (parameters.cc:285-330, defaults of the reference)
    surface_tension: float = 1.0
    gravity: float = 0.0
    epsilon: float = 1.0
    concentration_subdivisions: int = 2
    interpolate_grad_onto_pressure: bool = False   # "grad pressure compatible"
    curvature_correction: bool = False             # "curvature correction" (parameters.cc:314)
    convection_stabilization: bool = False         # "convection stabilization" (parameters.cc:360,580)
    # solver section (parameters.cc "Solver": defaults of the reference)
    max_nl_iteration: int = 10
    tol_nl_iteration: float = 1e-6
    max_lin_iteration: int = 500
    iterations_before_inner_solvers: int = 50        # "lin its before inner solvers", parameters.cc:212-223
    tol_lin_iteration: float = 1e-3
    rel_lin_iteration: bool = True
    # level-set driver (parameters.cc:351-357) and mesh keys a driver needs
    n_reinit_steps: int = 2
    n_initial_reinit_steps: int = 0
    global_refinements: int = 1
    adaptive_refinements: int = 0

    def __post_init__(self):
        if self.velocity_degree <= 1:
            raise NotImplementedError("velocity degree must be > 1 (parameters.cc:461-462)")
        if self.physical_type not in PHYSICAL_TYPES:
            raise NotImplementedError(self.physical_type)
        if self.linearization not in LINEARIZATIONS:
            raise ValueError("Linearization %s not available" % self.linearization)
        if self.physical_type == "incompressible stationary" and \
                self.linearization != "coupled implicit Newton":
            raise ValueError("Only coupled implicit Newton linearization available for "
                             "stationary equation")
        if self.tau_grad_div < 0:
            raise ValueError("Invalid parameter value")
        if self.physical_type == "stokes":  # parameters.cc:477-478
            self.density = 0.0
        # parameters.cc:593-595 (applied at parse time in the reference; here also for parameter
        # objects built in code, so that both behave like an input file with the same keys)
        if self.time_step_size_min > self.time_step_size_start:
            self.time_step_size_max = self.time_step_size_min = self.time_step_size_start

    @property
    def beta(self):
        return BETA[self.formulation_convective_term]

    @property
    def stored_damping(self):
        return -self.damping  # parameters.cc:466-467


# ---------------------------------------------------------------------------------------------
# deal.II ParameterHandler input files (the reference's tests/*.prm): `subsection X` ... `end`
# blocks, `set key = value` entries, `#` comments.  FlowParameters::parse_parameters
# (source/parameters.cc:449-614) reads the sections Navier-Stokes (+ its sub-section Solver),
# Two phase, Time stepping and Output options; the mapping below follows it line by line for the
# entries this engine uses.  Keys the engine has no use for (output, preconditioner names, AMR)
# are accepted and kept in `FlowParameters.unused`.
_PRM_KEYS = {
    ("Navier-Stokes", "dimension"): ("dimension", int),
    ("Navier-Stokes", "global refinements"): ("global_refinements", int),
    ("Navier-Stokes", "adaptive refinements"): ("adaptive_refinements", int),
    ("Navier-Stokes", "velocity degree"): ("velocity_degree", int),
    ("Navier-Stokes", "augmented Taylor-Hood elements"): ("augmented_taylor_hood", lambda v: int(v) > 0),
    ("Navier-Stokes", "viscosity"): ("viscosity", float),
    ("Navier-Stokes", "density"): ("density", float),
    ("Navier-Stokes", "damping"): ("damping", float),
    ("Navier-Stokes", "physical type"): ("physical_type", str),
    ("Navier-Stokes", "formulation convective term momentum balance"): ("formulation_convective_term", str),
    ("Navier-Stokes/Solver", "NL max iterations"): ("max_nl_iteration", int),
    ("Navier-Stokes/Solver", "NL tolerance"): ("tol_nl_iteration", float),
    ("Navier-Stokes/Solver", "linearization scheme"): ("linearization", str),
    ("Navier-Stokes/Solver", "tau grad div"): ("tau_grad_div", float),
    ("Navier-Stokes/Solver", "lin max iterations"): ("max_lin_iteration", int),
    ("Navier-Stokes/Solver", "lin its before inner solvers"): ("iterations_before_inner_solvers", int),
    ("Navier-Stokes/Solver", "lin tolerance"): ("tol_lin_iteration", float),
    ("Navier-Stokes/Solver", "lin relative tolerance"): ("rel_lin_iteration", lambda v: int(v) > 0),
    ("Two phase", "density difference"): ("density_diff", float),
    ("Two phase", "viscosity difference"): ("viscosity_diff", float),
    ("Two phase", "surface tension"): ("surface_tension", float),
    ("Two phase", "gravity"): ("gravity", float),
    ("Two phase", "epsilon"): ("epsilon", float),
    ("Two phase", "concentration subdivisions"): ("concentration_subdivisions", int),
    ("Two phase", "curvature correction"): ("curvature_correction", lambda v: int(v) > 0),
    ("Two phase", "convection stabilization"): ("convection_stabilization", lambda v: int(v) > 0),
    ("Two phase", "grad pressure compatible"): ("interpolate_grad_onto_pressure", lambda v: int(v) > 0),
    ("Two phase", "number reinit steps"): ("n_reinit_steps", int),
    ("Two phase", "number initial reinit steps"): ("n_initial_reinit_steps", int),
    ("Time stepping", "start time"): ("start_time", float),
    ("Time stepping", "end time"): ("end_time", float),
    ("Time stepping", "step size"): ("time_step_size_start", float),
    ("Time stepping", "max step size"): ("time_step_size_max", float),
    ("Time stepping", "min step size"): ("time_step_size_min", float),
    ("Time stepping", "scheme"): ("time_step_scheme", str),
}


def parse_prm(text):
    """{(section path, key): value string} of a ParameterHandler input text"""
    entries, path = {}, []
    for number, raw in enumerate(text.splitlines(), 1):
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        if line.startswith("subsection"):
            path.append(line[len("subsection"):].strip())
        elif line == "end":
            if not path:
                raise ValueError("line %d: 'end' without subsection" % number)
            path.pop()
        elif line.startswith("set"):
            if "=" not in line:
                raise ValueError("line %d: expected 'set key = value'" % number)
            key, value = line[3:].split("=", 1)
            entries[("/".join(path), " ".join(key.split()))] = value.strip()
        else:
            raise ValueError("line %d: cannot parse %r" % (number, raw))
    if path:
        raise ValueError("unclosed subsection %s" % path[-1])
    return entries


def flow_parameters_from_prm(text):
    """FlowParameters from the text of a .prm file (FlowParameters::parse_parameters)"""
    entries = parse_prm(text)
    kw, unused = {}, {}
    for (section, key), value in entries.items():
        target = _PRM_KEYS.get((section, key))
        if target is None:
            unused[(section, key)] = value
        else:
            kw[target[0]] = target[1](value)
    # parameters.cc:548-557: the Two phase section overrides density / viscosity when positive
    for name in ("density", "viscosity"):
        v = entries.get(("Two phase", name))
        if v is not None and float(v) > 0.0:
            kw[name] = float(v)
    p = FlowParameters(**kw)
    p.unused = unused
    return p
