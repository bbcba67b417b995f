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
    time_step_size_start: float = 0.05
    time_step_size_max: float = 1e10
    time_step_size_min: float = 0.0
    # two-phase section (parameters.cc:285-330, defaults of the reference)
    surface_tension: float = 1.0
    gravity: float = 0.0
    epsilon: float = 1.0
    concentration_subdivisions: int = 2
    interpolate_grad_onto_pressure: bool = False   # "grad pressure compatible"
    curvature_correction: bool = False             # "curvature correction" (parameters.cc:314)
    # solver section (parameters.cc "Solver": defaults of the reference)
    max_nl_iteration: int = 10
    tol_nl_iteration: float = 1e-6
    max_lin_iteration: int = 500
    tol_lin_iteration: float = 1e-3
    rel_lin_iteration: bool = True

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

    @property
    def beta(self):
        return BETA[self.formulation_convective_term]

    @property
    def stored_damping(self):
        return -self.damping  # parameters.cc:466-467
