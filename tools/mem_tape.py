"""Where does the device memory of a c3 step go?  Runs the generator forward + loss of one step under saved-tensor hooks and reports the
bytes autograd keeps alive for the backward, grouped by the Python frame that created the saving node (unique storages only), plus the
allocator's peak at the points of the step.  python tools/mem_tape.py [--config c3]"""
import argparse
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ag2video_amd.engine import TrainStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c3')
    ap.add_argument('--top', type=int, default=40)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    opt, side, frames, batch = bench.make_opt(args.config)
    opt.gpu_ids = [0]
    ts = TrainStep(opt, dev, weights_seed=0)
    clip = ts.to_device(bench.make_clip(opt, args.config, batch, frames, side, seed=1234))
    ts.gd_step(clip)
    torch.cuda.synchronize()
    gb = 1.0 / 2 ** 30
    print("after warm-up step: allocated %.1f GB, peak %.1f GB" % (torch.cuda.memory_allocated() * gb, torch.cuda.max_memory_allocated() * gb))
    base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()

    seen = {}
    by_site = collections.Counter()
    by_shape = collections.Counter()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def pack(t):
        if t.is_cuda:
            st = t.untyped_storage()
            key = st.data_ptr()
            if key not in seen:
                seen[key] = st.nbytes()
                site = '?'
                for fr in reversed(traceback.extract_stack(limit=40)):
                    if fr.filename.startswith(root) and '/models/' in fr.filename:
                        site = "%s:%d %s" % (os.path.relpath(fr.filename, root), fr.lineno, fr.name)
                        break
                by_site[site] += st.nbytes()
                by_shape[(tuple(t.shape), str(t.dtype).replace('torch.', ''))] += st.nbytes()
        return t

    imgs, objs, boxes, triplets, actions, video_id = clip
    with torch.autograd.graph.saved_tensors_hooks(pack, lambda t: t):
        model_out = ts.model(imgs, objs, triplets, actions, boxes_gt=boxes, test_mode=False, use_gt=True)
        torch.cuda.synchronize()
        g_fwd = sum(seen.values())
        print("generator forward: tape %.1f GB, allocated %.1f GB (+%.1f over the resident state)" % (
            g_fwd * gb, torch.cuda.memory_allocated() * gb, (torch.cuda.memory_allocated() - base) * gb))
        G_losses = ts.gans_model(clip, model_out, mode="compute_generator_loss")
        torch.cuda.synchronize()
        print("generator loss: tape %.1f GB, allocated %.1f GB, peak %.1f GB" % (
            sum(seen.values()) * gb, torch.cuda.memory_allocated() * gb, torch.cuda.max_memory_allocated() * gb))
    print("resident state (parameters, Adam, buffers, clip, caches): %.1f GB" % (base * gb))
    print("\n-- by creating frame --")
    for site, b in by_site.most_common(args.top):
        print("%8.2f GB  %s" % (b * gb, site))
    print("\n-- by shape --")
    for (shape, dt), b in by_shape.most_common(args.top):
        print("%8.2f GB  %s %s" % (b * gb, shape, dt))
    total = sum(v.mean() for v in G_losses.values()) if 'total_loss' not in G_losses else G_losses['total_loss'].mean()
    ts.optimizer_generator.zero_grad()
    total.backward()
    torch.cuda.synchronize()
    print("after backward: allocated %.1f GB, peak %.1f GB" % (torch.cuda.memory_allocated() * gb, torch.cuda.max_memory_allocated() * gb))


if __name__ == '__main__':
    main()
