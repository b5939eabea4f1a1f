"""Golden vectors for the PPO loop of the imitation task, recorded by RUNNING THE REFERENCE'S OWN METHODS:

  embodied_pose/agents/im_agent.py        ImitatorAgent.get_action_values (:271-294), _eval_critic (:296-303), play_steps (:305-409),
                                          prepare_dataset (:411-459), _calc_advs (:461-473), calc_gradients (:475-587)
  embodied_pose/learning/common_agent.py  discount_values (:423-435), bound_loss (:442-450), _actor_loss (:491-505), _critic_loss (:507-520)
  embodied_pose/models/im_models.py       ImitatorModel.Network.forward (:20-58)
  embodied_pose/models/im_network_builder.py   ImitatorBuilder.Network.forward_context / obtain_cur_context / preprocess_input /
                                          eval_actor (residual action, :226-228) / eval_critic / forward (:125-245)
  embodied_pose/models/running_norm.py    RunningNorm (training-mode update per minibatch, :32-43)

as unbound methods on a gym-less, rl_games-less stand-in for `self` (the technique of gen_golden_ball.py).  The "environment" is the
recorded reference trace tests/golden/env_trace.npz (6 envs, the first 32 steps of its epoch 0: observations, rewards, sticky resets,
terminations, context window), replayed step by step; the networks are the reference's own `Network` class with small MLPs
(734 -> 32 -> 16) whose weights are stored in the fixture.  rl-games itself (pinned 1.1.4, third-party, absent) is represented by
oracle/ref_shim/rl_games_restated.py - see its header for what that means for parity.

TEST INFRASTRUCTURE; runs in the build container only:
    PYTORCH_JIT=0 PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_ppo.py        -> tests/golden/ppo_trace.npz
"""
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
from torch import optim  # noqa: E402

import agents.im_agent as A  # noqa: E402
import models.im_models as IM  # noqa: E402
from utils.tools import AverageMeter  # noqa: E402

Agent = A.ImitatorAgent
torch.manual_seed(1234)
GOLD = os.path.join(HERE, "..", "tests", "golden")
tr = np.load(os.path.join(GOLD, "env_trace.npz"))
ops = np.load(os.path.join(GOLD, "task_ops.npz"))
N, T, PAD = 6, 32, 8
UNITS = (32, 16)
out = {"units": np.asarray(UNITS)}


def t32(x):
    return torch.tensor(np.asarray(x), dtype=torch.float32)


# ------------------------------------------------------------------ the recorded rollout as a scripted vec env
obs_seq = [t32(tr["e0_reset_obs"])] + [t32(tr["e0_s%02d_obs" % k]) for k in range(T)]
rew_seq = [t32(tr["e0_s%02d_rew" % k]) for k in range(T)]
done_seq = [torch.tensor(tr["e0_s%02d_reset" % k]).long() for k in range(T)]
term_seq = [torch.tensor(tr["e0_s%02d_terminate" % k]).long() for k in range(T)]
sub_seq = [t32(tr["e0_s%02d_sub_rewards" % k]) for k in range(T)]
context_feat = t32(tr["e0_context_feat"])
context_mask = torch.tensor(tr["e0_context_mask"]).bool()


# ------------------------------------------------------------------ the reference's Network with small MLPs
from ref_shim.ref_network import build_reference_network  # noqa: E402

net = build_reference_network(UNITS, PAD, ops)
model = IM.ImitatorModel.Network(net)
for k, v in model.state_dict().items():
    out["w0/" + k] = v.detach().numpy().copy()

value_mean_std = RG.RunningMeanStd((1,))
value_mean_std.running_mean[:] = 0.4
value_mean_std.running_var[:] = 2.5
value_mean_std.count.fill_(777.0)
out["vms0"] = np.array([0.4, 2.5, 777.0])


class Recorder:
    def __init__(self):
        self.items = []

    def update(self, x):
        self.items.append(x.detach().clone().reshape(-1))


def swap_and_flatten01(arr):  # (unused by the imitation agent: it keeps [N, T, ...] through swap01)
    raise AssertionError


# ------------------------------------------------------------------ stand-in for `self`
me = types.SimpleNamespace()
me.model, me.value_mean_std, me.normalize_value, me.normalize_input = model, value_mean_std, True, False
me.has_central_value, me.use_action_masks, me.is_rnn, me.rnn_states = False, False, False, None
me.horizon_length, me.num_agents, me.num_actors, me.batch_size = T, 1, N, N * T
me.gamma, me.tau, me.end_value_type = 0.99, 0.95, "next"
me.device = me.ppo_device = "cpu"
me.update_list = ["actions", "neglogpacs", "values", "mus", "sigmas"]
me.tensor_list = me.update_list + ["obses", "states", "dones", "next_obses"]
me.dones = torch.zeros(N, dtype=torch.uint8)
me.current_rewards, me.current_lengths = torch.zeros((N, 1)), torch.zeros(N)
me.game_rewards, me.game_lengths = Recorder(), Recorder()
me.algo_observer = types.SimpleNamespace(process_infos=lambda infos, ids: None, after_steps=lambda: None)
me.rewards_shaper = lambda r: r  # scale_value 1
me.task = types.SimpleNamespace(context_feat=context_feat, context_mask=context_mask)
me._preproc_obs = lambda o: o    # normalize_input False
f = dict(dtype=torch.float32)
me.experience_buffer = RG.ExperienceBuffer({
    "obses": torch.zeros((T, N, 461), **f), "next_obses": torch.zeros((T, N, 461), **f), "dones": torch.zeros((T, N), dtype=torch.uint8),
    "rewards": torch.zeros((T, N, 1), **f), "values": torch.zeros((T, N, 1), **f), "next_values": torch.zeros((T, N, 1), **f),
    "actions": torch.zeros((T, N, 75), **f), "neglogpacs": torch.zeros((T, N), **f), "mus": torch.zeros((T, N, 75), **f),
    "sigmas": torch.zeros((T, N, 75), **f)})


def set_eval():
    model.eval()
    value_mean_std.eval()


def set_train():
    model.train()
    value_mean_std.train()


me.set_eval, me.set_train = set_eval, set_train
step_counter = {"k": 0}


def env_reset():
    # VecTaskPythonWrapper.reset -> task.reset(): the task hands the context window to the registered model (humanoid_smpl_im.py:559-563)
    step_counter["k"] = 0
    with torch.no_grad():
        net.forward_context(context_feat, context_mask)
    return {"obs": obs_seq[0].clone()}


def env_step(actions):
    k = step_counter["k"]
    step_counter["k"] = k + 1
    infos = {"terminate": term_seq[k].clone(), "sub_rewards": sub_seq[k].clone(), "sub_rewards_names": "dof_rot,body_pos,body_rot,dof_vel"}
    return {"obs": obs_seq[k + 1].clone()}, rew_seq[k].clone().unsqueeze(1), done_seq[k].clone(), infos  # (A2CBase.env_step: rewards [N,1])


me.env_reset, me.env_step = env_reset, env_step
for name in ("get_action_values", "_eval_critic", "discount_values", "_calc_advs", "bound_loss", "_actor_loss", "_critic_loss"):
    setattr(me, name, types.MethodType(getattr(Agent, name), me))

# ------------------------------------------------------------------ 1. play_steps
with torch.no_grad():
    batch = Agent.play_steps(me)
for k in ("obses", "next_obses", "dones", "values", "actions", "neglogpacs", "mus", "sigmas", "returns", "alive"):
    out["play/" + k] = batch[k].numpy().copy()
td = me.experience_buffer.tensor_dict
out["play/next_values"] = td["next_values"].transpose(0, 1).numpy().copy()
out["play/rewards"] = td["rewards"].transpose(0, 1).numpy().copy()
out["play/played_frames"] = np.int64(batch["played_frames"])
out["play/alive_ratio"] = np.float64(me.alive_ratio)
out["play/game_rewards"] = torch.cat(me.game_rewards.items).numpy()
out["play/game_lengths"] = torch.cat(me.game_lengths.items).numpy()
out["play/step_rewards_avg"] = me.step_rewards.avg.numpy().astype(np.float64)
out["play/step_sub_rewards_avg"] = me.step_sub_rewards.avg.numpy().astype(np.float64)
out["play/step_count"] = np.float64(me.step_rewards.count)
out["env/rewards"] = torch.stack(rew_seq).numpy()
out["env/dones"] = torch.stack(done_seq).numpy()
out["env/terminate"] = torch.stack(term_seq).numpy()
out["env/sub_rewards"] = torch.stack(sub_seq).numpy()
out["env/obs"] = torch.stack(obs_seq).numpy()

# ------------------------------------------------------------------ 2. _calc_advs on its own (unnormalised values, as play_steps left them)
me.normalize_advantage = True
out["advs/normalized"] = Agent._calc_advs(me, batch).numpy().copy()
me.normalize_advantage = False
out["advs/raw"] = Agent._calc_advs(me, batch).numpy().copy()
me.normalize_advantage = True

# ------------------------------------------------------------------ 3. prepare_dataset (train mode: the value normaliser is updated with the
#        values, then with the returns, and normalises each with the statistics it has at that moment)
me.dataset = types.SimpleNamespace(update_values_dict=lambda d: setattr(me.dataset, "values_dict", d), values_dict=None)
me.set_train()
batch.pop("played_frames")
Agent.prepare_dataset(me, batch)
ds = me.dataset.values_dict
for k in ("old_values", "returns", "advantages", "old_logp_actions"):
    out["data/" + k] = ds[k].numpy().copy()
out["vms1"] = np.array([float(value_mean_std.running_mean), float(value_mean_std.running_var), float(value_mean_std.count)])

# ------------------------------------------------------------------ 4. calc_gradients: two mini-epochs x two minibatches of 3 envs
me.e_clip, me.critic_coef, me.entropy_coef, me.bounds_loss_coef, me.clip_value = 0.2, 5.0, 0.0, None, False
me.mixed_precision, me.multi_gpu, me.truncate_grads = False, False, True
me.last_lr = 1e-3  # (amass_im.yaml: 2e-5; larger here so that four updates move the weights well above float32 resolution)
me.optimizer = optim.Adam(model.parameters(), float(me.last_lr), eps=1e-08, weight_decay=0.0)
me.scaler = torch.cuda.amp.GradScaler(enabled=False)
special = ()
perms = [[4, 0, 3, 1, 5, 2], [2, 5, 1, 0, 3, 4]]
out["grad/perms"] = np.asarray(perms)
out["grad/lr"] = np.float64(me.last_lr)
call = 0
for ep, perm in enumerate(perms):
    for i in range(2):
        idx = torch.tensor(perm[3 * i:3 * i + 3])
        me.grad_norm = 50.0 if call != 2 else 0.05  # (the third update runs into the gradient-norm clip)
        inp = {k: v[idx] for k, v in ds.items() if v is not None}  # AMPDataset._get_item (learning/amp_datasets.py:17-31)
        Agent.calc_gradients(me, inp)
        r = me.train_result
        for k in ("actor_loss", "critic_loss", "entropy", "kl", "actor_clip_frac"):
            out["grad/%d/%s" % (call, k)] = np.float64(float(r[k]))
        out["grad/%d/grad_norm_clip" % call] = np.float64(me.grad_norm)
        out["grad/%d/rn_mean" % call] = net.running_obs.mean.numpy().copy()
        out["grad/%d/rn_std" % call] = net.running_obs.std.numpy().copy()
        out["grad/%d/rn_n" % call] = np.int64(int(net.running_obs.n))
        for k, v in model.state_dict().items():
            if "running_obs" not in k:
                out["grad/%d/w/%s" % (call, k)] = v.detach().numpy().copy()
        call += 1
out["grad/calls"] = np.int64(call)

np.savez_compressed(os.path.join(GOLD, "ppo_trace.npz"), **out)
print("wrote tests/golden/ppo_trace.npz: %d arrays" % len(out))
for k in ("play/alive_ratio", "play/step_rewards_avg", "play/game_lengths", "vms1", "grad/0/actor_loss", "grad/0/critic_loss", "grad/0/kl", "grad/2/actor_clip_frac", "grad/3/rn_n"):
    print(" ", k, out[k])
