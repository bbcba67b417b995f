"""Host mirror of adaflo::TimeStepping (source/time_stepping.cc:23-200): the BDF-2 /
implicit-Euler weights and extrapolation factors the kernels take as scalars."""


class TimeStepping:
    def __init__(self, parameters):
        p = parameters
        self.start_val = p.start_time
        self.final_val = p.end_time
        self.scheme = p.time_step_scheme
        self.start_step_val = p.time_step_size_start
        self.max_step_val = p.time_step_size_max
        self.min_step_val = p.time_step_size_min
        self.current_step_val = p.time_step_size_start
        self.last_step_val = 0.0
        self.step_val = p.time_step_size_start
        self.weight_val = 1.0 / p.time_step_size_start
        self.weight_old_val = -1.0
        self.weight_old_old_val = 0.0
        self.factor_extrapol_old = 0.0
        self.factor_extrapol_old_old = 0.0
        self.step_no_val = 0
        self.at_end_val = False
        self.weight_changed = True
        self.now_val = self.start_val
        self.prev_val = self.start_val
        self.tau1_val, self.tau2_val = {"implicit_euler": (1.0, 0.0), "explicit_euler": (0.0, 1.0),
                                        "crank_nicolson": (0.5, 0.5), "bdf_2": (1.0, 0.0)}[self.scheme]

    # accessors, include/adaflo/time_stepping.h:230-294
    def weight(self): return self.weight_val
    def weight_old(self): return self.weight_old_val
    def weight_old_old(self): return self.weight_old_old_val
    def tau1(self): return self.tau1_val
    def tau2(self): return self.tau2_val
    def step_size(self): return self.current_step_val
    def old_step_size(self): return self.last_step_val
    def now(self): return self.now_val
    def previous(self): return self.prev_val
    def step_no(self): return self.step_no_val
    def at_end(self): return self.at_end_val

    def extrapolate(self, old, old_old):
        return old * self.factor_extrapol_old + old_old * self.factor_extrapol_old_old

    def set_time_step(self, value):
        self.current_step_val = value
        self.step_val = value

    def set_desired_time_step(self, desired_value):
        """source/time_stepping.cc:247-268: at most a factor 2 away from the previous step size and within
        [min step size, max step size]"""
        prev = desired_value if self.now_val == 0 else self.current_step_val
        step = min(2 * prev, max(desired_value, 0.5 * prev))
        self.current_step_val = min(self.max_step_val, max(self.min_step_val, step))
        self.step_val = self.current_step_val

    def restart(self):
        """source/time_stepping.cc:104-119"""
        self.step_no_val = 0
        self.now_val = self.start_val
        self.step_val = self.start_step_val
        self.current_step_val = self.step_val
        self.last_step_val = 0.0
        self.at_end_val = (self.final_val - self.start_val) / self.start_step_val < 1e-14
        self.weight_changed = True

    def name(self):
        return {"implicit_euler": "ImplEuler", "explicit_euler": "ExplEuler", "crank_nicolson": "CrankNicolson",
                "bdf_2": "BDF-2"}[self.scheme]

    def at_tick(self, tick):
        """source/time_stepping.cc:225-235"""
        time = self.now_val
        slot = int(time * 1.0000000001 / tick) * tick
        return not ((time - slot) > self.current_step_val * 0.95 and not self.at_end_val)

    def next(self):
        """source/time_stepping.cc:123-200"""
        assert not self.at_end_val, "Final time already reached, cannot proceed"
        s = self.current_step_val
        if self.now_val != self.start_val:
            self.last_step_val = self.current_step_val
            if self.scheme == "bdf_2" and self.step_no_val == 1:
                s = self.step_val
            if s > self.max_step_val:
                s = self.max_step_val
        h = self.now_val + s
        self.current_step_val = s
        s1 = 0.01 * s
        if not self.at_end_val and h > self.final_val - s1:
            self.current_step_val = self.final_val - self.now_val
            h = self.final_val
            self.at_end_val = True
        c, l = self.current_step_val, self.last_step_val
        if self.scheme == "bdf_2" and self.now_val != self.start_val:
            new_weight = (2.0 * c + l) / (c * (c + l))
            self.weight_old_val = -((c + l) / (c * l))
            self.weight_old_old_val = c / (l * (c + l))
        else:
            new_weight = 1.0 / c
            self.weight_old_val = -1.0 / c
        if abs(new_weight - self.weight_val) / new_weight > 1e-12:
            self.weight_val = new_weight
            self.weight_changed = True
        else:
            self.weight_changed = False
        if self.step_no_val > 1:
            self.factor_extrapol_old = (c + l) / l
            self.factor_extrapol_old_old = -c / l
        else:
            self.factor_extrapol_old = 1.0
            self.factor_extrapol_old_old = 0.0
        self.prev_val = self.now_val
        self.now_val = h
        self.step_no_val += 1
        return self.now_val
