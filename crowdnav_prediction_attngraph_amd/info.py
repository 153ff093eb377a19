"""Episode-info objects, mirroring crowd_sim/envs/utils/info.py (same class names and __str__ so that
rl/evaluation.py-style isinstance checks and logs keep working)."""


class Timeout(object):
    def __str__(self):
        return "Timeout"


class ReachGoal(object):
    def __str__(self):
        return "Reaching goal"


class Danger(object):
    def __init__(self, min_dist):
        self.min_dist = min_dist

    def __str__(self):
        return "Too close"


class Collision(object):
    def __str__(self):
        return "Collision"


class Nothing(object):
    def __str__(self):
        return ""


_NOTHING, _TIMEOUT, _COLLISION, _REACHGOAL = Nothing(), Timeout(), Collision(), ReachGoal()


def from_code(code, min_dist=0):
    """CN_INFO_* -> info object (train phase: Danger.min_dist is 0, crowd_sim_var_num.py:496-498; test phase: the distance
    to the closest intruded future position, :499-511)."""
    if code == 0:
        return _NOTHING
    if code == 1:
        return _TIMEOUT
    if code == 2:
        return _COLLISION
    if code == 3:
        return _REACHGOAL
    if code == 4:
        return Danger(min_dist)
    raise ValueError("unknown info code %r" % (code,))
