"""CPU: the host logic of the online launch-policy tuner (gym-lmaze_amd/vec_env.py OnlineTuner), with stand-in
events -- no GPU.  Importing the package loads liblmaze_hip.so (cross-compiled here), which needs no device."""
import importlib

PKG = importlib.import_module("gym-lmaze_amd")
vec_env = importlib.import_module("gym-lmaze_amd.vec_env")


class FakeEvent:
    """An event pair 'completes' only once the test says so; elapsed_time returns what the test planted."""

    def __init__(self, ms=None):
        self.ms, self.done = ms, False

    def query(self):
        return self.done

    def elapsed_time(self, other):
        return other.ms


def test_round_robin_warm_up_and_lowest_median_wins():
    cands = ((3, 1), (3, 2), (8, 1))
    cost = {(3, 1): 0.090, (3, 2): 0.083, (8, 1): 0.100}
    t = vec_env.OnlineTuner(cands, warm=5, samples=3)
    pending, best, seen = [], None, []
    for i in range(200):
        c = t.next_candidate()
        seen.append(c)
        e0, e1 = FakeEvent(), FakeEvent(cost[c] * (3.0 if i % 7 == 0 else 1.0))   # an outlier now and then
        pending.append(e1)
        if len(pending) > 4:                 # the device runs a few launches behind the host
            pending.pop(0).done = True
        best = t.add(c, e0, e1)
        if best is not None:
            break
    assert seen[:6] == [cands[0], cands[1], cands[2]] * 2           # strict round robin
    assert best == (3, 2)                                            # medians shrug the outliers off
    assert all(len(v) >= 3 for v in t.timings.values())
    assert i >= 5 + 3 * len(cands) - 1                               # never before every candidate has its samples


def test_nothing_is_decided_while_the_device_lags():
    t = vec_env.OnlineTuner(((3, 1), (8, 1)), warm=0, samples=2)
    evs = []
    for i in range(50):                      # no event ever completes
        c = t.next_candidate()
        e0, e1 = FakeEvent(), FakeEvent(0.1)
        evs.append(e1)
        assert t.add(c, e0, e1) is None
    for e in evs:
        e.done = True
    c = t.next_candidate()
    assert t.add(c, FakeEvent(), FakeEvent(0.1)) in ((3, 1), (8, 1))


def test_launch_hint_encoding():
    V = PKG.LmazeVecEnv
    assert V.launch_hint_of(3, 2) == 0x23 and V.launch_hint_of(8) == 0x18 and V.launch_hint_of(0, 0) == 0
    # (0, 0) = launch_hint 0, the library's per-shape default: always a candidate, and the one kept unless beaten by 1.5 %
    assert V.CANDIDATES[0] == V.DEFAULT_POLICY == (0, 0) and (3, 1) in V.CANDIDATES and (3, 2) in V.CANDIDATES
    assert all(1 <= c[0] <= 8 and 1 <= c[1] <= 15 and (len(c) == 2 or c[2] in (1, 2, 3)) for c in V.CANDIDATES[1:])
    assert V.launch_hint_of(8, 1, 2) == 0x818 and V.launch_hint_of(4, 1, 1) == 0x414
