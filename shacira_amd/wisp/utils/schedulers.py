"""Hyper-parameter decay schedules with the reference's formulas (wisp/utils/schedulers.py:4-31): used by the
trainers for the entropy weight (cosine) and the SGA temperature (exp)."""
import math


class DecayScheduler(object):
    def __init__(self, total_steps, decay_name="fix", start=0, end=0, params=None):
        self.decay_name, self.start, self.end = decay_name, start, end
        self.total_steps, self.params = total_steps, params

    def __call__(self, step):
        kind, a, b, n = self.decay_name, self.start, self.end, self.total_steps
        if kind == "fix":
            return a
        if kind == "linear":
            return b if step > n else a + (b - a) * step / n
        if kind == "exp":
            rate = -math.log(1 / self.params["temperature"]) * step / n / self.params["decay_period"]
            return max(b, a * math.exp(rate))
        if kind == "inv_sqrt":
            return a * (n / (n + step)) ** 0.5
        if kind == "cosine":
            return b + 0.5 * (a - b) * (1 + math.cos(step / n * math.pi))
        raise ValueError("Unknown decay name: {}".format(kind))
