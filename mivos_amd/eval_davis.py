"""The DAVIS interactive-track evaluation loop on the MI355X engine: what the reference's entry script
`eval_interactive_davis.py:22-108` does, as a function and a CLI with the script's own arguments.

    python -m mivos_amd.eval_davis --prop_model saves/propagation_model.pth --fusion_model saves/fusion.pth \
           --s2m_model saves/s2m.pth --davis ../DAVIS/2017 --output out --save_mask

The reference script itself also runs unchanged on the engine (`python -m mivos_amd.dropin eval_interactive_davis.py ...`,
INTEGRATION.md); this module exists for deployments without the reference tree and differs from the script in two places only:
the clips are decoded once and ingested straight into HBM by HIP kernels (`dataset.DAVISTestDataset(device=...)`, no DataLoader
workers, no host-side float tensors), and the session class is injectable (`session_factory`).  Everything else - one fresh
DAVISProcessor per (sequence, user) sample, `processor.interact(scribbles)` / `sess.submit_masks(pred_masks, next_masks)`, the
masks of a sample written as palette PNGs to `<output>/<user_iter>/<sequence>/` when the NEXT sample starts (so the last sample
is never written, :84-99), `summary.json` - follows the script line by line, so both produce the same files."""
import argparse
import os
from os import path

import torch

from . import clip_io
from .dataset.davis_test_dataset import DAVISTestDataset
from .davis_processor import DAVISProcessor
from .model.fusion_net import FusionNet
from .model.propagation.prop_net import PropagationNetwork
from .model.s2m.s2m_network import deeplabv3plus_resnet50 as S2M


def run_interactive_davis(davis_path, out_path, prop_model, fusion_model, s2m_model, save_mask=False, device="cuda:0", session_factory=None,
                          max_nb_interactions=8, max_time=8 * 30, report_save_dir="../output", log=print):
    """prop_model / fusion_model / s2m_model: state dicts or checkpoint paths (eval_interactive_davis.py:56-68).  Returns
    (report, summary) of the session."""
    from PIL import Image
    if session_factory is None:
        from davisinteractive.session.session import DavisInteractiveSession as session_factory
    os.makedirs(out_path, exist_ok=True)
    palette = Image.open(path.expanduser(davis_path + "/trainval/Annotations/480p/blackswan/00000.png")).getpalette()      # :38
    torch.autograd.set_grad_enabled(False)
    dataset = DAVISTestDataset(davis_path + "/trainval", imset="2017/val.txt", device=device)
    images, num_objects = {}, {}
    for i in range(len(dataset)):                                                      # :47-53 ("loads all the images")
        data = dataset[i]
        name = data["info"]["name"]
        images[name] = data["rgb"].unsqueeze(0)                                         # [1,T,3,H,W], resident in HBM
        num_objects[name] = len(data["info"]["labels"])
    log("Finished loading %d sequences." % len(images))

    def load(net, state):
        net.load_state_dict(torch.load(state, map_location="cpu") if isinstance(state, (str, os.PathLike)) else state)
        return net.to(device).eval()
    prop, fuse, s2m = load(PropagationNetwork(), prop_model), load(FusionNet(), fusion_model), load(S2M(), s2m_model)

    total_iter, user_iter, last_seq, pred_masks, processor = 0, 0, None, None, None
    with session_factory(davis_root=davis_path + "/trainval", report_save_dir=report_save_dir, max_nb_interactions=max_nb_interactions,
                         max_time=max_time) as sess:
        while sess.next():
            sequence, scribbles, new_seq = sess.get_scribbles(only_last=True)
            if new_seq:
                processor = None                                                        # every pre-computed feature of the sample goes (:80-81)
                processor = DAVISProcessor(prop, fuse, s2m, images[sequence], num_objects[sequence], device=device)
                log(sequence)
                if save_mask:                                                           # "save last time" (:86-99)
                    if pred_masks is not None:
                        clip_io.write_palette_png(pred_masks, palette, path.join(out_path, str(user_iter), last_seq))
                    if last_seq is None or sequence != last_seq:
                        last_seq, user_iter = sequence, 0
                    else:
                        user_iter += 1
            pred_masks, next_masks, this_idx = processor.interact(scribbles)
            sess.submit_masks(pred_masks, next_masks)
            total_iter += 1
        report = sess.get_report()
        summary = sess.get_global_summary(save_file=path.join(out_path, "summary.json"))
    return report, summary


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--prop_model", default="saves/propagation_model.pth")
    ap.add_argument("--fusion_model", default="saves/fusion.pth")
    ap.add_argument("--s2m_model", default="saves/s2m.pth")
    ap.add_argument("--davis", default="../DAVIS/2017")
    ap.add_argument("--output")
    ap.add_argument("--save_mask", action="store_true")
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    run_interactive_davis(a.davis, a.output, a.prop_model, a.fusion_model, a.s2m_model, save_mask=a.save_mask, device=a.device)


if __name__ == "__main__":
    main()
