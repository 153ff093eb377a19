"""LazyInfos: the `infos` value of BatchedCrowdSim.step() must behave like the list of E dicts the reference's vec-env returns
(rl/networks/shmem_vec_env.py:136-142 + bench.Monitor: {'info': obj} per env, plus 'episode' for envs that finished) while building
dicts only for finished envs.  CPU-only: no device involved."""
import numpy as np
import pytest

from crowdnav_prediction_attngraph_amd import info as I


def _mk(E=300, seed=0, with_danger=False):
    rs = np.random.RandomState(seed)
    codes = rs.choice([0, 0, 0, 0, 1, 2, 3] + ([4] if with_danger else []), size=E).astype(np.uint8)
    done = np.isin(codes, (1, 2, 3))
    ep_ret, ep_len = rs.uniform(-20, 10, E), rs.randint(1, 200, E).astype(np.int32)
    md = rs.uniform(0.3, 1.0, E) if with_danger else None
    return codes, done, ep_ret, ep_len, md, I.LazyInfos(codes, np.flatnonzero(done), ep_ret, ep_len, 12.5, md)


def _eager(codes, done, ep_ret, ep_len, md):
    out = []
    for i, c in enumerate(codes):
        d = {"info": I.from_code(int(c), float(md[i]) if (md is not None and c == 4) else 0)}
        if done[i]:
            d["episode"] = {"r": round(float(ep_ret[i]), 6), "l": int(ep_len[i]), "t": 12.5}
        out.append(d)
    return out


@pytest.mark.parametrize("with_danger", [False, True])
def test_lazy_infos_equal_the_eager_list(with_danger):
    codes, done, ep_ret, ep_len, md, lazy = _mk(with_danger=with_danger)
    want = _eager(codes, done, ep_ret, ep_len, md)
    assert len(lazy) == len(want)
    for views in (list(lazy), [lazy[i] for i in range(len(lazy))], lazy[:], [lazy[i - len(lazy)] for i in range(len(lazy))]):
        for got, w in zip(views, want):
            assert set(got.keys()) == set(w.keys())
            assert type(got["info"]) is type(w["info"]) and str(got["info"]) == str(w["info"])
            if isinstance(w["info"], I.Danger):
                assert got["info"].min_dist == w["info"].min_dist
            if "episode" in w:
                assert got["episode"] == w["episode"]
    # the loops train.py:180-189 runs over it
    rewards = [info["episode"]["r"] for info in lazy if "episode" in info.keys()]
    assert rewards == [w["episode"]["r"] for w in want if "episode" in w]
    assert all("bad_transition" not in info.keys() for info in lazy)
    with pytest.raises(IndexError):
        lazy[len(lazy)]


def test_lazy_infos_build_dicts_only_for_finished_envs_and_shared_ones_are_read_only():
    codes, done, _, _, _, lazy = _mk(E=4096, seed=1)
    running = np.flatnonzero(~done)
    a, b = lazy[int(running[0])], lazy[int(running[-1])]
    assert a is b or codes[running[0]] != codes[running[-1]], "unfinished envs of one info code share one dict"
    assert len(lazy._own) == int(done.sum())
    with pytest.raises(TypeError):
        a["episode"] = {}
    d = lazy[int(np.flatnonzero(done)[0])]
    d["extra"] = 1                      # a finished env's dict is its own
    assert lazy[int(np.flatnonzero(done)[0])]["extra"] == 1


def test_lazy_infos_copy_deepcopy_and_pickle_like_the_reference_list():
    """Wrappers in the gym / baselines style deepcopy, pickle or annotate the infos they are handed (the reference returns E fresh dicts):
    the copies must be plain writable dicts and the shared originals must stay untouched."""
    import copy
    import pickle
    codes, done, ep_ret, ep_len, md, lazy = _mk(E=64, seed=3, with_danger=True)
    want = _eager(codes, done, ep_ret, ep_len, md)
    for clone in (copy.deepcopy(list(lazy)), pickle.loads(pickle.dumps(list(lazy))), [copy.copy(d) for d in lazy], [d.copy() for d in lazy]):
        assert len(clone) == len(want)
        for got, w in zip(clone, want):
            assert type(got) is dict and set(got) == set(w)
            assert type(got["info"]) is type(w["info"]) and got.get("episode") == w.get("episode")
        for got in clone:
            got["bad_transition"] = True            # writable, and private to the copy (deepcopy / pickle keep ONE copy per shared original)
    assert all("bad_transition" not in d for d in lazy)
    # the LazyInfos object itself survives a deepcopy / pickle round trip as a list-like of the same content
    for clone in (copy.deepcopy(lazy), pickle.loads(pickle.dumps(lazy))):
        assert [str(d["info"]) for d in clone] == [str(w["info"]) for w in want]
