"""Golden vectors for the evaluation rollout (`run.py --test`), recorded by RUNNING THE REFERENCE'S OWN METHODS:

  embodied_pose/players/im_player.py   ImitatorPlayer.get_action (:143-167), env_step (:169-186), run (:192-311)

as unbound methods on a gym-less, rl_games-less stand-in for `self` (the technique of gen_golden_ppo.py; the same small reference
`Network`, its weights taken from tests/golden/ppo_trace.npz).  The "environment" is the recorded reference trace
tests/golden/env_trace.npz replayed step by step: the 36 steps of its epoch 0, then the 4 steps of its epoch 1 - 40 steps, so that the
player runs past the 32-step context window and calls `task._init_context(task._reset_ref_motion_ids, task._cur_ref_motion_times)` at
n = 32 (the scripted task then shows the window of epoch 1).  What rl_games' BasePlayer [1.1.4, third-party, absent] would supply -
`get_batch_size`, `_preproc_obs` with normalize_input False, the player options - is set on the stand-in by hand.

Two scenarios, both deterministic (the sampled path draws from torch's global CPU generator, which no device kernel can follow; the
sampling arithmetic itself is pinned by the PPO goldens):
  a  games_num 4: envs finish at steps 9, 20, 33, 34 -> the round ends when the fourth game is counted;
  b  envs 0 and 2 swapped, games_num 3: env 0 finishes at step 9 -> `if done[0]: break` ends every round there, the env is reset and the
     next round starts.

TEST INFRASTRUCTURE; runs in the build container only:
    PYTORCH_JIT=0 PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_player.py        -> tests/golden/player_trace.npz
"""
import contextlib
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
os.environ.setdefault("PYTORCH_JIT", "0")
sys.dont_write_bytecode = True

from ref_shim import install as shim  # noqa: E402

shim.install()
from ref_shim import rl_games_restated as RG  # noqa: E402

RG.register()

import torch  # noqa: E402

import models.im_models as IM  # noqa: E402
import players.im_player as PL  # noqa: E402
from ref_shim.ref_network import build_reference_network  # noqa: E402

Player = PL.ImitatorPlayer
GOLD = os.path.join(HERE, "..", "tests", "golden")
tr = np.load(os.path.join(GOLD, "env_trace.npz"))
ops = np.load(os.path.join(GOLD, "task_ops.npz"))
ppo = np.load(os.path.join(GOLD, "ppo_trace.npz"))
N, PAD, CTX = 6, 8, 32
UNITS = tuple(int(u) for u in ppo["units"])


def t32(x):
    return torch.tensor(np.asarray(x), dtype=torch.float32)


torch.manual_seed(4321)
net = build_reference_network(UNITS, PAD, ops)
model = IM.ImitatorModel.Network(net)
model.load_state_dict({k[3:]: torch.tensor(ppo[k]) for k in ppo.keys() if k.startswith("w0/")})
model.eval()

# ------------------------------------------------------------------ the recorded rollout: 36 steps of epoch 0, then 4 of epoch 1
steps = [("e0", k) for k in range(36)] + [("e1", k) for k in range(4)]
obs_seq = [t32(tr["e0_reset_obs"])] + [t32(tr["%s_s%02d_obs" % s]) for s in steps]
rew_seq = [t32(tr["%s_s%02d_rew" % s]) for s in steps]
# sticky like the task's reset_buf within one round: once done, done
done_seq, acc = [], torch.zeros(N, dtype=torch.long)
for s in steps:
    acc = torch.maximum(acc, torch.tensor(tr["%s_s%02d_reset" % s]).long())
    done_seq.append(acc.clone())
windows = [(t32(tr["e%d_context_feat" % e]), torch.tensor(tr["e%d_context_mask" % e]).bool()) for e in (0, 1)]
out = {"num_steps": np.int64(len(steps)), "env/obs": torch.stack(obs_seq).numpy(), "env/rewards": torch.stack(rew_seq).numpy(),
       "env/dones": torch.stack(done_seq).numpy(), "env/context_feat": torch.stack([w[0] for w in windows]).numpy(),
       "env/context_mask": torch.stack([w[1] for w in windows]).numpy()}


def scenario(perm, games_num):
    perm_t = torch.tensor(perm)
    state = {"k": 0, "resets": 0, "context_calls": [], "actions": []}

    def show(window):
        with torch.no_grad():
            net.forward_context(windows[window][0][perm_t], windows[window][1][perm_t])

    task = types.SimpleNamespace(context_length=CTX, _reset_ref_motion_ids=torch.arange(N), _cur_ref_motion_times=torch.zeros(N))
    task.render_vis = lambda init=False: None

    def init_context(motion_ids, motion_times):
        assert motion_ids is task._reset_ref_motion_ids and motion_times is task._cur_ref_motion_times
        state["context_calls"].append(state["k"])
        show(1)

    task._init_context = init_context

    class Env:  # (no has_action_mask / create_agent attributes: the player asks for them with getattr)
        def step(self, actions):
            k = state["k"]
            state["k"] = k + 1
            state["actions"].append(actions.detach().clone())
            return {"obs": obs_seq[k + 1][perm_t].clone()}["obs"], rew_seq[k][perm_t].clone(), done_seq[k][perm_t].clone(), {}

    me = types.SimpleNamespace()
    me.env, me.task, me.model = Env(), task, model
    me.games_num, me.render_env, me.n_game_life, me.is_determenistic = games_num, False, 1, True
    me.is_rnn, me.states, me.rnn_states, me.has_batch_dimension = False, None, None, True
    me.device, me.max_steps, me.num_agents, me.print_stats, me.render_sleep = "cpu", len(steps), 1, True, 0.0
    me.clip_actions, me.is_tensor_obses, me.value_size = False, True, 1
    me._preproc_obs = lambda o: o        # BasePlayer._preproc_obs with normalize_input False
    me._post_step = lambda info: None     # CommonPlayer._post_step

    def get_batch_size(obses, batch_size):  # BasePlayer.get_batch_size: a batched observation tensor
        return obses.size()[0]

    def env_reset(env_ids=None):          # CommonPlayer.env_reset -> task.reset(): the task hands the window to the registered model
        state["k"] = 0
        state["resets"] += 1
        show(0)
        return {"obs": obs_seq[0][perm_t].clone()}

    me.get_batch_size, me.env_reset = get_batch_size, env_reset

    def env_step(env, actions):  # ImitatorPlayer.env_step hands back {'obs': obs}
        return Player.env_step(me, env, actions)

    me.env_step = env_step
    me.get_action = types.MethodType(Player.get_action, me)
    text = io.StringIO()
    with contextlib.redirect_stdout(text):
        Player.run(me)
    lines = text.getvalue().strip().splitlines()
    per_game = [(float(ln.split()[1]), float(ln.split()[3])) for ln in lines if ln.startswith("reward:")]
    total = float(lines[-2])
    last = lines[-1].split()
    return {"actions": torch.stack(state["actions"]).numpy(), "per_step_stats": np.asarray(per_game, dtype=np.float64), "sum_rewards": np.float64(total),
            "av_reward": np.float64(last[2]), "av_steps": np.float64(last[5]), "resets": np.int64(state["resets"]),
            "context_calls": np.asarray(state["context_calls"], dtype=np.int64), "perm": np.asarray(perm, dtype=np.int64), "games_num": np.int64(games_num)}


for tag, perm, games in (("a", [0, 1, 2, 3, 4, 5], 4), ("b", [2, 1, 0, 3, 4, 5], 3)):
    res = scenario(perm, games)
    for k, v in res.items():
        out["%s/%s" % (tag, k)] = v
    print(tag, "env steps", res["actions"].shape[0], "resets", int(res["resets"]), "context rebuilt at", res["context_calls"].tolist(),
          "sum", float(res["sum_rewards"]), "av reward", float(res["av_reward"]), "av steps", float(res["av_steps"]))

np.savez_compressed(os.path.join(GOLD, "player_trace.npz"), **out)
print("wrote tests/golden/player_trace.npz:", len(out), "arrays")
