"""Behavioural acceptance tests of the shield — the only expectations about shield OUTCOMES the reference states itself
(demos/demo_gym_functionality_Schunk_fullstop_criterion.py:36-86 and ..._pfl_criterion.py:1-8,36-86):
a static T-pose human next to the arm, the arm driven by action[:6] = clip(goal - q, -1, 1) with
goal = [1.4 sin(8 pi t / 200), 1.5, 0, 0, 0, 0], control_freq 5.  "With PFL, robot can move past the human with reduced
speed.  With SSM, robot cannot move past the human because it is too close."""
import numpy as np
import pytest

import human_robot_gym_amd as hrg

STEPS = 200


def _kw(shield):
    return dict(shield_type=shield, control_freq=5, horizon=1000, done_at_success=False, done_at_collision=False)


def _clips():
    return hrg.static_clip(600, pelvis=(0.0, 1.0, 0.95))  # T-pose, 0.95 m in front of the robot, arms along +-y


def _drive(B, get_q, step):
    q1, pv, safe = [], [], []
    info = None
    for t in range(STEPS):
        q = get_q()
        goal = np.array([1.4 * np.sin(t * np.pi / 200 * 8), 1.5, 0, 0, 0, 0])
        a = np.zeros((1, 7))
        a[0, :6] = np.clip(goal - q, -1, 1)
        info = step(a)
        s = B.get_state(0)
        q1.append(s.qpos[0]); pv.append(s.path_v); safe.append(s.is_safe)
    return np.array(q1), np.array(pv), np.array(safe), info


def _oracle_run(shield):
    from oracle.oracle import OracleBatch
    clips = _clips()
    B = OracleBatch(hrg.build_model_desc(_kw(shield), n_clips=1, goal_check=False), clips, 1)
    B.reset()
    return _drive(B, lambda: np.array(B.get_state(0).qpos[:6]), lambda a: B.step(a)[3][0])


def test_ssm_cannot_pass_the_human():
    q1, pv, safe, info = _oracle_run("SSM")
    assert info[2] == 0                                   # no collision at all
    assert info[8] > 100                                  # failsafe_interventions on most steps
    assert q1.min() > -0.05                               # the sweep to the far side never happens
    assert np.ptp(q1[-80:]) < 1e-3 and pv[-1] == 0.0      # parked in front of the human, fully stopped
    assert safe[-80:].sum() == 0


def test_pfl_passes_at_reduced_speed():
    d = hrg.build_model_desc(_kw("PFL"), n_clips=1)
    q1, pv, safe, info = _oracle_run("PFL")
    assert d.failsafe_sdot == 0.0 and d.pfl_v_safe == 0.25 and all(0.1 < r < 1.5 for r in d.pfl_reach)
    assert q1.min() < -0.2                                # it does get past the (kinematic, immovable: D1) human's arm, which parks SSM at q1 > -0.05: the sweep
                                                          # ploughs on through the contact; how far it gets in the 200 steps is contact dynamics, not shield logic
    assert pv[20:].min() > 0.02                           # never a full stop
    assert info[8] > 100 and info[11] == 0
    unsafe = ~safe.astype(bool)
    assert unsafe.mean() > 0.5 and np.median(pv[unsafe][20:]) < 0.6   # while the reach sets intersect the fast sweep is throttled


def _slow_run(B, step):
    pv, safe, vq = [], [], []
    for t in range(STEPS):
        q = np.array(B.get_state(0).qpos[:6])
        goal = np.array([1.4 * np.sin(t * np.pi / 200 * 8), 1.5, 0, 0, 0, 0])
        a = np.zeros((1, 7))
        a[0, :6] = np.clip(goal - q, -0.15, 0.15)          # x 0.2 rad per unit action: at most 0.03 rad per policy step
        step(a)
        s = B.get_state(0)
        pv.append(s.path_v); safe.append(s.is_safe); vq.append(max(abs(v) for v in s.qvel[:6]))
    return np.array(pv), ~np.array(safe, bool), np.array(vq)


def test_pfl_lets_a_slow_approach_keep_its_own_speed():
    """The same scene approached slowly (goal steps of 0.03 rad: the planned trajectory never moves a point of the arm faster than pfl_v_safe): the PFL shield does
    not slow it any further -- the path speed stays 1 although the reachable sets intersect; the constant-fraction model of round 1 throttled it to ~ 4 mm/s."""
    from oracle.oracle import OracleBatch
    clips = _clips()
    B = OracleBatch(hrg.build_model_desc(_kw("PFL"), n_clips=1, goal_check=False), clips, 1)
    B.reset()
    pv, unsafe, vq = _slow_run(B, lambda a: B.step(a))
    assert unsafe.mean() > 0.3                              # the human is within reach for much of the run
    assert pv.min() > 0.95                                  # ... and the arm is hardly slowed below its own (slow) speed: with the joints of a trajectory moving
                                                            # together (time synchronisation) their point speeds add up to 1.6 % over pfl_v_safe at the peak
    assert max(vq) < 0.35                                   # which is slow indeed: joint speeds stay below 0.35 rad/s
    B.close()


@pytest.mark.gpu
def test_hip_lets_a_slow_approach_keep_its_own_speed():
    import torch
    from human_robot_gym_amd._lib import HipBatch
    G = HipBatch(hrg.build_model_desc(_kw("PFL"), n_clips=1, goal_check=False), _clips(), 1)
    G.reset()
    pv, unsafe, vq = _slow_run(G, lambda a: G.step(torch.from_numpy(a).cuda()))
    assert unsafe.mean() > 0.3 and pv.min() > 0.95 and max(vq) < 0.35
    G.close()


def test_off_runs_into_the_human():
    q1, pv, safe, info = _oracle_run("OFF")
    assert info[2] > 50 and info[8] == 0 and pv.min() == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("shield", ["SSM", "PFL"])
def test_hip_reproduces_the_scenario(shield):
    import torch
    from human_robot_gym_amd._lib import HipBatch
    clips = _clips()
    G = HipBatch(hrg.build_model_desc(_kw(shield), n_clips=1, goal_check=False), clips, 1)
    G.reset()
    q1g, pvg, sg, ig = _drive(G, lambda: np.array(G.get_state(0).qpos[:6]), lambda a: G.step(torch.from_numpy(a).cuda())[3].cpu().numpy()[0])
    q1o, pvo, so, io = _oracle_run(shield)
    if shield == "SSM":                                   # contact-free scenario: trajectories agree to tolerance
        np.testing.assert_allclose(q1g, q1o, rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(sg, so)
        np.testing.assert_array_equal(ig, io)
    else:                                                 # PFL touches the human (contact dynamics are chaotic): same verdicts
        assert q1g.min() < -0.2 and pvg[20:].min() >= pvo[20:].min() * (1 - 1e-5) - 1e-9 and ig[11] == 0   # (the slowest path speed: within the parity tolerance of the oracle's)
        k = int(np.argmax(np.abs(q1g - q1o) > 1e-4)) if (np.abs(q1g - q1o) > 1e-4).any() else STEPS
        assert k > 20                                      # identical until well into the first contact phase
    G.close()
