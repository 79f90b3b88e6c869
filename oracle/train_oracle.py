"""CPU restatement of the FusionNet training step (reference `model/fusion_model.py:54-131`).

TEST INFRASTRUCTURE ONLY (see oracle/stm_oracle.py's header).  Functional torch + autograd on a flat ``state_dict``:
``stm_oracle.fusion_net`` (fusion_net.py:32-50) on both objects, ``aggregate_wbg_channel`` (aggregate.py:39-53),
BootstrappedCE / LossComputer (losses.py:21-60), ``torch.optim.Adam(lr, weight_decay=1e-7)`` (fusion_model.py:44-45).
PIN: ``tests/test_oracle_golden.py::test_train_step_oracle_matches_the_reference`` compares it with
``tests/golden/train_small.npz``, which ``oracle/make_golden_train.py`` produced by running the unmodified reference modules.
"""
import torch
import torch.nn.functional as F

from . import stm_oracle as O


def bootstrapped_ce(logits, target, it, start_warm, end_warm, top_p=0.15):
    """losses.py:21-41."""
    if it < start_warm:
        return F.cross_entropy(logits, target), 1.0
    raw = F.cross_entropy(logits, target, reduction="none").view(-1)
    if it > end_warm:
        this_p = top_p
    else:
        this_p = top_p + (1 - top_p) * ((end_warm - it) / (end_warm - start_warm))
    loss, _ = torch.topk(raw, int(raw.numel() * this_p), sorted=False)
    return loss.mean(), this_p


def train_step(fsd, data, attn1, attn2, it, iterations, lr, dtype=torch.float32):
    """One do_pass in train mode.  fsd: FusionNet state_dict; data: the batch dict of dataset/fusion_dataset.py:225-249; attn1 /
    attn2: the (no-grad) attention maps of fusion_model.py:81-82.  Returns logits, mask, total_loss, p, grads and the parameters
    after one Adam step."""
    params = {k: v.detach().clone().to(dtype).requires_grad_(True) for k, v in fsd.items()}
    d = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in data.items()}
    prob1 = torch.sigmoid(O.fusion_net(params, d["rgb"], d["seg1"], d["seg2"], attn1.to(dtype), d["dist"]))
    prob2 = torch.sigmoid(O.fusion_net(params, d["rgb"], d["seg12"], d["seg22"], attn2.to(dtype), d["dist"]))
    prob = torch.cat([prob1, prob2], 1) * d["selector"].unsqueeze(2).unsqueeze(2)             # fusion_model.py:86
    logits, mask = O.aggregate_wbg_channel(prob, True)
    b = d["gt"].shape[0]
    start_warm, end_warm = int(iterations * 0.2), int(iterations * 0.5)                          # losses.py:47
    total, p_sum = 0, 0
    for j in range(b):                                                                          # losses.py:55-63
        if d["selector"][j][1] > 0.5:
            loss, p = bootstrapped_ce(logits[j:j + 1], d["cls_gt"][j:j + 1], it, start_warm, end_warm)
        else:
            loss, p = bootstrapped_ce(logits[j:j + 1, :2], d["cls_gt"][j:j + 1], it, start_warm, end_warm)
        total = total + loss / b
        p_sum += p / b
    names = list(params)
    opt = torch.optim.Adam([params[n] for n in names], lr=lr, weight_decay=1e-7)
    opt.zero_grad(set_to_none=True)
    total.backward()
    grads = {n: params[n].grad.detach().clone() for n in names}
    opt.step()
    return dict(logits=logits.detach(), mask=mask.detach(), total_loss=float(total.item()), p=float(p_sum), grads=grads,
                new={n: params[n].detach().clone() for n in names})
