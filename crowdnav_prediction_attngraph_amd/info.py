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


class _SharedInfo(dict):
    """{'info': obj} handed out for MANY envs at once (LazyInfos): read-only, so that one caller's write cannot show up in another env's
    info.  The reference only reads these dicts (train.py:180-182, 188-189; rl/evaluation.py:118-140)."""

    def _ro(self, *a, **k):
        raise TypeError("this info dict is shared between envs and read-only; copy it (dict(info)) to modify")

    __setitem__ = __delitem__ = update = pop = popitem = clear = setdefault = _ro

    # copies and pickles are PLAIN dicts (writable, private to whoever made them): gym / baselines-style wrappers deepcopy, pickle or
    # annotate the infos list they are handed, which the reference's list of E fresh dicts allows
    def __copy__(self):
        return dict(self)

    def copy(self):
        return dict(self)

    def __deepcopy__(self, memo):
        import copy
        return {k: copy.deepcopy(v, memo) for k, v in self.items()}

    def __reduce__(self):
        return (dict, (dict(self),))


class LazyInfos(object):
    """The `infos` list of VecEnv.step() (list of E dicts, shmem_vec_env.py:136-142 + bench.Monitor) without E dict constructions per step:
    an env that did not finish shares the one read-only {'info': obj} of its info code, only finished envs (and Danger in the test phase,
    which carries a per-env distance) get a dict of their own, built when first asked for.  Indexing, len(), iteration, `in` and slicing
    behave like the list the reference returns; list(infos) materialises it."""

    __slots__ = ("_codes", "_own", "_min_dist")
    _SHARED = None

    def __init__(self, codes, done_idx, ep_ret, ep_len, now, min_dist=None):
        if LazyInfos._SHARED is None:
            LazyInfos._SHARED = [_SharedInfo(info=from_code(c)) for c in range(4)]
        self._codes = codes.tolist()                      # info code per env (python ints: cheap to index)
        self._min_dist = min_dist
        self._own = {}
        for i in done_idx:
            i = int(i)
            self._own[i] = {"info": from_code(self._codes[i]), "episode": {"r": round(float(ep_ret[i]), 6), "l": int(ep_len[i]), "t": now}}

    def __len__(self):
        return len(self._codes)

    def _get(self, i):
        d = self._own.get(i)
        if d is not None:
            return d
        c = self._codes[i]
        if c == 4:      # Danger: per-env min_dist (0 in the train phase, crowd_sim_var_num.py:496-498)
            d = self._own[i] = {"info": Danger(float(self._min_dist[i]) if self._min_dist is not None else 0)}
            return d
        return LazyInfos._SHARED[c]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._get(j) for j in range(*i.indices(len(self._codes)))]
        n = len(self._codes)
        if i < 0:
            i += n
        if not 0 <= i < n:
            raise IndexError("info index out of range")
        return self._get(i)

    def __iter__(self):
        own, shared, get = self._own, LazyInfos._SHARED, self._get
        for i, c in enumerate(self._codes):
            yield shared[c] if (c < 4 and i not in own) else get(i)
