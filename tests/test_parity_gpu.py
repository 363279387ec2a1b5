"""HIP stepper vs CPU oracle on identical seeded inputs (through the C ABI).  -m gpu."""
import numpy as np
import pytest

from helpers import ATOL, RTOL, assert_state_close, make_pair

pytestmark = pytest.mark.gpu


def _rollout(kw, n_envs, n_steps, seed, resync):
    import torch
    O, G = make_pair(n_envs, kw)
    oo = O.reset()
    og = G.reset().cpu().numpy()
    np.testing.assert_allclose(og, oo, rtol=RTOL, atol=ATOL)
    rng = np.random.RandomState(seed)
    for k in range(n_steps):
        a = rng.uniform(-1, 1, (n_envs, 7))
        o_o, r_o, d_o, i_o = O.step(a)
        o_g, r_g, d_g, i_g = G.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        msg = f"step {k}"
        np.testing.assert_array_equal(i_g.cpu().numpy(), i_o, err_msg=msg)
        np.testing.assert_array_equal(d_g.cpu().numpy(), d_o, err_msg=msg)
        np.testing.assert_allclose(o_g.cpu().numpy(), o_o, rtol=RTOL, atol=1e-6, err_msg=msg)
        np.testing.assert_allclose(r_g.cpu().numpy(), r_o, rtol=RTOL, atol=1e-6, err_msg=msg)
        np.testing.assert_allclose(G.term_obs.cpu().numpy(), O.term_obs, rtol=RTOL, atol=1e-6, err_msg=msg)
        po, no = O.contacts()
        pg, ng = G.contacts()
        np.testing.assert_array_equal(ng, no, err_msg=msg)   # contact-pair indices bit-exact
        np.testing.assert_array_equal(pg, po, err_msg=msg)
        for e in range(n_envs):
            so, sg = O.get_state(e), G.get_state(e)
            assert_state_close(so, sg, f"{msg} env {e}")
            if resync:
                G.set_state(e, so)
    O.close(); G.close()


@pytest.mark.parametrize("shield", ["OFF", "SSM"])
def test_step_parity_resync(shield):
    """Per-step parity with the GPU state re-synchronised to the oracle after every step."""
    kw = dict(shield_type=shield, reward_shaping=True, base_human_pos_offset=[1.0, 0.0, 0.0], horizon=20)
    _rollout(kw, n_envs=16, n_steps=45, seed=1, resync=True)


@pytest.mark.parametrize("shield", ["OFF", "SSM"])
def test_step_parity_free_running(shield):
    """Free-running rollouts (no resync) incl. auto-resets: trajectories stay within tolerance."""
    kw = dict(shield_type=shield, reward_shaping=True, base_human_pos_offset=[0.9, 0.1, 0.0], human_rand=[0.3, 0.3, 0.5], horizon=15)
    _rollout(kw, n_envs=32, n_steps=40, seed=2, resync=False)


def test_contacts_and_collisions_occur():
    """Human standing inside the robot's workspace: contacts are detected, classified and agree."""
    kw = dict(shield_type="OFF", base_human_pos_offset=[0.45, 0.0, 0.0], horizon=30, done_at_collision=False, collision_reward=-10)
    import torch
    O, G = make_pair(16, kw)
    O.reset(); G.reset()
    rng = np.random.RandomState(5)
    tot = 0
    for k in range(30):
        a = rng.uniform(-1, 1, (16, 7))
        a[:, 1] = 1.0  # drive joint 2 forward into the human / table
        _, _, _, i_o = O.step(a)
        _, _, _, i_g = G.step(torch.from_numpy(a).cuda())
        np.testing.assert_array_equal(i_g.cpu().numpy(), i_o, err_msg=f"step {k}")
        po, no = O.contacts(); pg, ng = G.contacts()
        np.testing.assert_array_equal(ng, no); np.testing.assert_array_equal(pg, po)
        tot += int(i_o[:, 2].sum())
        for e in range(16):
            assert_state_close(O.get_state(e), G.get_state(e), f"step {k} env {e}")
    assert tot > 0, "scenario was meant to produce collisions"
