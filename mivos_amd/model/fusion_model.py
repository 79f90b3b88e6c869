"""FusionModel for MI355X - the training step of the difference-aware FusionNet (reference `model/fusion_model.py:17-131`,
`model/losses.py:21-76`, `train.py:96-124`): same constructor arguments, ``do_pass(data, it)``, ``train() / val() / test()``,
``save / load_model / load_network / load_prop``.

What runs where:
  * attention maps: `AttentionReadNetwork` (frozen, no gradient - fusion_model.py:81-82), the engine's inference kernels;
  * FusionNet forward on both objects as ONE batch of 2B samples (the two calls share the weights, so their gradients add),
    layer by layer on `mivos_conv2d_fused` with the activations kept for the backward pass;
  * loss: `mivos_fusion_loss` (sigmoid x selector -> aggregate_wbg_channel -> per-pixel cross-entropy), BootstrappedCE's
    top-p selection by `mivos_fusion_kth_loss` (exact radix select per sample);
  * backward: `mivos_fusion_loss_grad`; data gradients = 3x3 convolutions with the transposed / rotated weights on the
    same convolution kernels (residual adds in their epilogues), ReLU masks by `mivos_mul_positive`, weight / bias
    gradients by `mivos_fusion_wgrad3x3` (exact fp32 MFMA, deterministic summation order);
  * data parallelism like the reference's DistributedDataParallel: one process per GPU, ONE all-reduce (RCCL over xGMI) of the
    flat 39 905-element gradient per step, averaged over the ranks; parameters are views of one flat vector;
  * `mivos_adam_step`: torch.optim.Adam(lr, weight_decay=1e-7) on the flat vector, MultiStepLR on the host.
PyTorch is plumbing here (tensor storage, the process group); no autograd graph is built.
"""
import math
import os

import torch

from .. import ops, shard
from .._lib import MivosHipError
from ..ops import ConvLayer
from .attn_network import AttentionReadNetwork
from .fusion_net import FusionNet


def _flatten_parameters(net):
    """Make every parameter of `net` a view of one flat fp32 vector (state_dict names / shapes unchanged)."""
    params = [p for p in net.parameters() if p.requires_grad]
    flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n] = p.data.reshape(-1)
        p.data = flat[off:off + n].view_as(p)
        off += n
    return flat, params


# ---- torch.optim.Adam / MultiStepLR state_dict layouts (what the reference's checkpoints hold, fusion_model.py:145-157) ----------

def adam_state_dict(shapes, exp_avg, exp_avg_sq, step, lr_now, lr_initial, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7):
    """The flat Adam moments as `torch.optim.Adam(...).state_dict()`: per-parameter state {step, exp_avg, exp_avg_sq} in the order
    of `filter(requires_grad, net.parameters())` (= named_parameters order, = the flat vector's order) + one param group."""
    state, off = {}, 0
    for i, shp in enumerate(shapes):
        n = 1
        for d in shp:
            n *= d
        if step > 0:                                   # torch creates a parameter's state at its first step()
            state[i] = dict(step=torch.tensor(float(step)), exp_avg=exp_avg[off:off + n].detach().cpu().reshape(shp).clone(),
                            exp_avg_sq=exp_avg_sq[off:off + n].detach().cpu().reshape(shp).clone())
        off += n
    group = dict(lr=float(lr_now), betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, foreach=None,
                 capturable=False, differentiable=False, fused=None, initial_lr=float(lr_initial), params=list(range(len(shapes))))
    return dict(state=state, param_groups=[group])


def flat_from_adam_state_dict(sd, shapes, device):
    """Inverse of adam_state_dict for checkpoints written by torch.optim.Adam (any torch version: `step` an int or a tensor) ->
    (exp_avg flat, exp_avg_sq flat, step)."""
    total = 0
    for shp in shapes:
        n = 1
        for d in shp:
            n *= d
        total += n
    m, v = torch.zeros(total, dtype=torch.float32, device=device), torch.zeros(total, dtype=torch.float32, device=device)
    if len(sd["param_groups"]) != 1 or len(sd["param_groups"][0]["params"]) != len(shapes):
        raise MivosHipError("optimizer state_dict does not describe FusionNet's %d parameters in one group" % len(shapes))
    step, off = 0, 0
    for i, shp in zip(sd["param_groups"][0]["params"], shapes):
        n = 1
        for d in shp:
            n *= d
        st = sd["state"].get(i)
        if st is not None:
            if tuple(st["exp_avg"].shape) != tuple(shp):
                raise MivosHipError(f"optimizer state of parameter {i} has shape {tuple(st['exp_avg'].shape)}, expected {tuple(shp)}")
            m[off:off + n] = st["exp_avg"].reshape(-1).to(device=device, dtype=torch.float32)
            v[off:off + n] = st["exp_avg_sq"].reshape(-1).to(device=device, dtype=torch.float32)
            step = max(step, int(round(float(st["step"]))))
        off += n
    return m, v, step


def multistep_state_dict(milestones, gamma, lr_initial, last_epoch):
    """`torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones, gamma).state_dict()` after `last_epoch` scheduler steps."""
    from collections import Counter
    lr = lr_initial * gamma ** sum(1 for m in milestones if m <= last_epoch)
    return dict(milestones=Counter(milestones), gamma=gamma, base_lrs=[float(lr_initial)], last_epoch=int(last_epoch), verbose=False,
                _step_count=int(last_epoch) + 1, _get_lr_called_within_step=False, _last_lr=[float(lr)])


def get_iou_hook(values):                   # losses.py:8-9
    return "iou/iou", (values["hide_iou/i"] + 1) / (values["hide_iou/u"] + 1)


def get_sec_iou_hook(values):               # losses.py:11-12
    return "iou/sec_iou", (values["hide_iou/sec_i"] + 1) / (values["hide_iou/sec_u"] + 1)


iou_hooks = [get_iou_hook, get_sec_iou_hook]


class Integrator:
    """Running means of the logged losses (reference util/log_integrator.py:10-78, the part FusionModel uses): add_dict per
    iteration, finalize(prefix, it) averages, sums over the ranks (one reduce of a scalar per key) and hands rank 0's logger
    `log_metrics(prefix, key, value, it)`."""

    def __init__(self, logger, distributed=False, local_rank=0, world_size=1):
        self.values, self.counts, self.hooks = {}, {}, []
        self.logger, self.distributed, self.local_rank, self.world_size = logger, distributed, local_rank, world_size

    def add_tensor(self, key, value):
        # device scalars stay on the device until finalize(): no host synchronisation per iteration
        v = value.detach().float().mean() if torch.is_tensor(value) else float(value)
        self.values[key] = v if key not in self.values else self.values[key] + v
        self.counts[key] = self.counts.get(key, 0) + 1

    def add_dict(self, d):
        for k, v in d.items():
            self.add_tensor(k, v)

    def add_hook(self, hook):
        self.hooks.extend(hook if isinstance(hook, list) else [hook])

    def reset_except_hooks(self):
        self.values, self.counts = {}, {}

    def finalize(self, prefix, it, f=None):
        for hook in self.hooks:
            k, v = hook(self.values)
            self.add_tensor(k, v)
        for k, v in self.values.items():
            if k[:4] == "hide":
                continue
            avg = float(v) / self.counts[k]
            if self.distributed:
                avg = shard.mean_over_ranks(avg)
            if self.logger is not None and self.local_rank == 0:
                self.logger.log_metrics(prefix, k, avg, it, f)


class BootstrappedSchedule:
    """losses.py:21-41: before start_warm the plain mean, then the mean of the top this_p fraction of the per-pixel losses."""

    def __init__(self, start_warm, end_warm, top_p=0.15):
        self.start_warm, self.end_warm, self.top_p = start_warm, end_warm, top_p

    def fraction(self, it):
        if it < self.start_warm:
            return None
        if it > self.end_warm:
            return self.top_p
        return self.top_p + (1 - self.top_p) * ((self.end_warm - it) / (self.end_warm - self.start_warm))


class FusionModel:
    def __init__(self, para, logger=None, save_path=None, local_rank=0, world_size=1, distributed=True):
        self.para, self.local_rank, self.world_size = para, local_rank, world_size
        self.distributed = distributed and world_size > 1
        if not torch.cuda.is_available():
            raise MivosHipError("FusionModel needs an MI355X; mivos_amd has no CPU execution path")
        self.device = torch.device("cuda", local_rank % torch.cuda.device_count())
        self.net = FusionNet().to(self.device)
        self.prop_net = AttentionReadNetwork().eval().to(self.device)
        self.flat, self.params = _flatten_parameters(self.net)
        if self.distributed:                               # DistributedDataParallel broadcasts rank 0's parameters at construction
            shard.broadcast_parameters(self.flat, 0)
        self.grad = torch.zeros_like(self.flat)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.lr, self.weight_decay, self.betas, self.eps = para["lr"], 1e-7, (0.9, 0.999), 1e-8
        self.milestones, self.gamma = list(para["steps"]), para["gamma"]
        self.opt_step = 0
        self.bce = BootstrappedSchedule(int(para["iterations"] * 0.2), int(para["iterations"] * 0.5))
        self.logger, self.save_path = logger, save_path
        # logging / checkpoint cadence of the reference (fusion_model.py:31-52)
        import time
        self.last_time = time.time()
        self.train_integrator = Integrator(logger, distributed=self.distributed, local_rank=local_rank, world_size=world_size)
        self.train_integrator.add_hook(iou_hooks)                               # fusion_model.py:38: train/iou/iou, train/iou/sec_iou
        self.val_integrator = Integrator(logger, distributed=self.distributed, local_rank=local_rank, world_size=world_size)
        self.report_interval, self.save_im_interval, self.save_model_interval = 100, 500, 5000
        if para.get("debug"):
            self.report_interval = self.save_im_interval = 1
        self.train()

    # ---- modes (fusion_model.py:203-224; BN-free networks: the flags only gate the backward pass / logging) --------------
    def train(self):
        self._is_train, self._do_log = True, True
        self.integrator = self.train_integrator
        return self

    def val(self):
        self._is_train, self._do_log = False, True
        self.integrator = self.val_integrator
        return self

    def finalize_val(self, it):
        """fusion_model.py:190-192: log and reset the validation means."""
        self.val_integrator.finalize("val", it)
        self.val_integrator.reset_except_hooks()

    def test(self):
        self._is_train, self._do_log = False, False
        return self

    def current_lr(self):
        """optim.lr_scheduler.MultiStepLR(milestones, gamma) after `opt_step` scheduler steps."""
        return self.lr * self.gamma ** sum(1 for m in self.milestones if m <= self.opt_step)

    # ---- forward with saved activations ----------------------------------------------------------------------------------
    def _forward(self, im, seg_a, seg_b, attn, dist_pair):
        """The two FusionNet calls of fusion_model.py:84-85 as one batch: samples [0, B) = object 1, [B, 2B) = object 2."""
        B, _, H, W = im.shape
        P = H * W
        x16 = torch.empty((2 * B, H, W, 16), dtype=torch.float32, device=self.device)
        tl = dist_pair.detach().float().cpu().tolist()
        for o in range(2):
            for b in range(B):
                planes, _, _ = FusionNet._planes((im[b], 0), (seg_a[o][b], 0), (seg_b[o][b], 0), (attn[o][b], 0), tl[b])
                x16[o * B + b] = ops.interleave(planes, 1, P, 16, self.device).view(H, W, 16)
        c1, c2a, c2b, c3a, c3b, fin = self.net.plan()
        x1 = ops.conv(x16, c1, relu_out=True)
        r1 = ops.conv(x1, c2a, relu_out=True)
        x2 = ops.conv(r1, c2b, res=x1, relu_out=True)
        r2 = ops.conv(x2, c3a, relu_out=True)
        x3 = ops.conv(r2, c3b, res=x2, relu_out=True)
        z = ops.fusion_head(x3, fin) if ops.CONV_PRECISION == "f16x3" else ops.conv(x3, fin)
        return z, (x16, x1, r1, x2, r2, x3)

    @staticmethod
    def _dgrad_layer(conv_params, cin_pad=None):
        """The convolution that maps d(output) to d(input) of a 3x3 / pad 1 / stride 1 convolution: weights transposed (in <-> out)
        and rotated by 180 degrees, no bias."""
        w = conv_params.weight.detach().flip(2, 3).permute(1, 0, 2, 3).contiguous()
        return ConvLayer.pack(w, None, None, 1, 1, cin_pad=cin_pad)

    def _backward(self, dz, acts):
        """dz [2B,H,W,1] = d loss / d logits -> self.grad (flat, the order of self.params)."""
        x16, x1, r1, x2, r2, x3 = acts
        S, H, W, _ = x16.shape
        net = self.net
        grads = {}

        def wgrad(name, x, g):
            dw, db = ops.fusion_wgrad3x3(x, g)                              # OHWI [cg,3,3,cx], [cg]
            grads[name] = (dw, db)

        wgrad("final_conv", x3, dz)
        g16 = ops.interleave([(dz.reshape(-1), H * W)], S, H * W, 16, self.device).view(S, H, W, 16)
        g3 = ops.conv(g16, self._dgrad_layer(net.final_conv, cin_pad=16).to(self.device))
        ops.mul_positive(g3, x3)                                            # x3 = relu(x2 + conv3b(r2))
        wgrad("conv3.2", r2, g3)
        gr2 = ops.conv(g3, self._dgrad_layer(net.conv3[2]).to(self.device))
        ops.mul_positive(gr2, r2)                                           # r2 = relu(conv3a(x2))
        wgrad("conv3.0", x2, gr2)
        g2 = ops.conv(gr2, self._dgrad_layer(net.conv3[0]).to(self.device), res=g3)      # + the identity path of the block
        ops.mul_positive(g2, x2)
        wgrad("conv2.2", r1, g2)
        gr1 = ops.conv(g2, self._dgrad_layer(net.conv2[2]).to(self.device))
        ops.mul_positive(gr1, r1)
        wgrad("conv2.0", x1, gr1)
        g1 = ops.conv(gr1, self._dgrad_layer(net.conv2[0]).to(self.device), res=g2)
        ops.mul_positive(g1, x1)
        wgrad("conv1.0", x16, g1)
        named = dict(net.named_parameters())
        off = 0
        for name, p in named.items():
            if not p.requires_grad:
                continue
            layer, kind = name.rsplit(".", 1)
            dw, db = grads[layer]
            if kind == "weight":
                g = dw[..., :p.shape[1]].permute(0, 3, 1, 2)              # OHWI (input channels padded) -> OIHW
            else:
                g = db
            self.grad[off:off + p.numel()] = g.reshape(-1)
            off += p.numel()
        return grads

    # ---- one iteration (fusion_model.py:54-131) ----------------------------------------------------------------------------
    def do_pass(self, data, it=0):
        with ops.on_device(self.device), torch.no_grad(), ops.single_stream_region():
            d = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in data.items()}
            im = d["rgb"].float().contiguous()
            B, _, H, W = im.shape
            P = H * W
            attn1, attn2 = self.prop_net(d["src2_ref_im"], d["src2_ref"], d["src2_ref_gt"], d["src2_ref2"], d["src2_ref_gt2"], im)
            self.net.refresh_plan_if_stale()
            z, acts = self._forward(im, (d["seg1"].float(), d["seg12"].float()), (d["seg2"].float(), d["seg22"].float()),
                                    (attn1, attn2), d["dist"])
            z = z.view(2, B, P)
            selector = d["selector"].float().contiguous()
            cls_gt = d["cls_gt"].to(torch.int32).contiguous().view(B, P)
            logits, mask, loss = ops.fusion_loss(z[0], z[1], selector, cls_gt)
            out = dict(logits=logits.view(B, 3, H, W), mask=mask.view(B, 3, H, W), attn1=attn1, attn2=attn2)
            losses = None
            if self._do_log or self._is_train:
                frac = self.bce.fraction(it)
                if frac is None:
                    k = P
                    per_sample = loss.view(B, P).sum(1) / P            # F.cross_entropy(reduction='mean')
                    wsel = torch.tensor([[-math.inf, 1.0 / (P * B), 0.0]] * B, dtype=torch.float32, device=self.device)
                    this_p = 1.0
                else:
                    k = int(P * frac)
                    kk = torch.full((B,), k, dtype=torch.int32, device=self.device)
                    sel = ops.fusion_kth_loss(loss, kk)                 # [B,4] = tau, #(> tau), sum(> tau), #(== tau)
                    tau, n_gt, s_gt, n_eq = sel[:, 0], sel[:, 1], sel[:, 2], sel[:, 3]
                    per_sample = (s_gt + (k - n_gt) * tau) / k           # mean of the k largest (ties have equal values)
                    wsel = torch.stack([tau, torch.full_like(tau, 1.0 / (k * B)), (k - n_gt) / n_eq / (k * B)], 1).contiguous()
                    this_p = frac
                losses = {"total_loss": per_sample.sum() / B, "p": this_p}
                if self._do_log:
                    # losses.py:66-73: intersection / union sums of the two objects' masks against their ground truth (a logging metric:
                    # two compares and four reductions per iteration, no host synchronisation before finalize)
                    m = out["mask"]
                    for tag, seg, gt in (("", m[:, 1:2] > 0.5, d["gt"] > 0.5), ("sec_", m[:, 2:3] > 0.5, d["gt2"] > 0.5)):
                        losses[f"hide_iou/{tag}i"] = (seg & gt).float().sum()
                        losses[f"hide_iou/{tag}u"] = (seg | gt).float().sum()
                    self.integrator.add_dict(losses)                   # (image dumps of the reference's logger, :99-113, are not reproduced)
            if self._is_train:
                import time
                if it % self.report_interval == 0 and it != 0:          # fusion_model.py:115-121
                    if self.logger is not None:
                        self.logger.log_scalar("train/lr", self.current_lr(), it)
                        self.logger.log_metrics("train", "time", (time.time() - self.last_time) / self.report_interval, it)
                    self.last_time = time.time()
                    self.train_integrator.finalize("train", it)
                    self.train_integrator.reset_except_hooks()
                if it % self.save_model_interval == 0 and it != 0 and self.logger is not None:      # :123-125: a job killed mid-run keeps
                    self.save(it)                                                                   # its last periodic checkpoint
                dz1, dz2 = ops.fusion_loss_grad(z[0], z[1], selector, cls_gt, loss, wsel)
                dz = torch.stack([dz1, dz2], 0).view(2 * B, H, W, 1)
                self._backward(dz, acts)
                if self.distributed:                                    # DistributedDataParallel: gradients averaged over the ranks
                    shard.average_gradients(self.grad)
                self.opt_step += 1
                ops.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.current_lr_for_step(), self.betas, self.eps,
                              self.weight_decay, self.opt_step)
                self.net.invalidate_plan()                              # the packed weights follow the updated parameters
            out["losses"] = losses
            return out

    def current_lr_for_step(self):
        # optimizer.step() of iteration n uses the rate the scheduler set after n - 1 scheduler steps
        return self.lr * self.gamma ** sum(1 for m in self.milestones if m <= self.opt_step - 1)

    # ---- checkpoints (fusion_model.py:133-195) -----------------------------------------------------------------------------
    def save(self, it):
        if self.save_path is None:
            print("Saving has been disabled.")
            return
        os.makedirs(os.path.dirname(self.save_path), exist_ok=True)
        torch.save(self.net.state_dict(), self.save_path + ("_%s.pth" % it))
        self.save_checkpoint(it)

    def save_checkpoint(self, it):
        if self.save_path is None:
            print("Saving has been disabled.")
            return
        os.makedirs(os.path.dirname(self.save_path), exist_ok=True)
        # the reference's layout (fusion_model.py:152-157): torch.optim.Adam / MultiStepLR state_dicts, so either side resumes the other's run
        shapes = [tuple(p.shape) for p in self.params]
        torch.save({"it": it, "network": {k: v.detach().cpu() for k, v in self.net.state_dict().items()},
                    "optimizer": adam_state_dict(shapes, self.exp_avg, self.exp_avg_sq, self.opt_step, self.current_lr(), self.lr, self.betas, self.eps,
                                                 self.weight_decay),
                    "scheduler": multistep_state_dict(self.milestones, self.gamma, self.lr, self.opt_step)}, self.save_path + "_checkpoint.pth")

    def _load_net_state(self, state):
        self.net.load_state_dict(state)
        self.flat, self.params = _flatten_parameters(self.net)
        self.net.invalidate_plan()

    def load_model(self, path):
        """Resume from a `*_checkpoint.pth` written by this class OR by the reference's FusionModel (same layout: 'network',
        torch.optim.Adam's 'optimizer' state_dict, MultiStepLR's 'scheduler' state_dict); the private layout of this class's
        round-3 checkpoints ({step, exp_avg, exp_avg_sq} flat) is still read."""
        # tensors, numbers, dicts and MultiStepLR's `collections.Counter` of milestones: loaded with the restricted unpickler
        from collections import Counter
        with torch.serialization.safe_globals([Counter]):
            ck = torch.load(path, map_location="cpu", weights_only=True)
        self._load_net_state(ck["network"])
        opt, sch = ck["optimizer"], ck.get("scheduler") or {}
        if "param_groups" in opt:
            self.exp_avg, self.exp_avg_sq, step = flat_from_adam_state_dict(opt, [tuple(p.shape) for p in self.params], self.device)
            self.opt_step = int(sch.get("last_epoch", step))
            if "base_lrs" in sch:
                self.lr = float(sch["base_lrs"][0])
            if "milestones" in sch:
                self.milestones, self.gamma = sorted(sch["milestones"].elements()) if hasattr(sch["milestones"], "elements") else list(sch["milestones"]), sch.get("gamma", self.gamma)
        else:
            self.opt_step = opt["step"]
            self.exp_avg, self.exp_avg_sq = opt["exp_avg"].to(self.device), opt["exp_avg_sq"].to(self.device)
        self.grad = torch.zeros_like(self.flat)
        return ck["it"]

    def load_network(self, path):
        self._load_net_state(torch.load(path, map_location=self.device))

    def load_prop(self, path):
        self.prop_net.load_state_dict(torch.load(path, map_location=self.device), strict=False)
