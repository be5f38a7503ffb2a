"""One rank of a data-parallel step on the HIP model: the real ``GradSync`` (per-bucket all-reduce on a side stream during backward,
embedding bucket deferred) + ``HipAdamW`` (everything but the deferred bucket first), against the single-process step over the union
of the micro-batches.  Launched by ``tests/test_dp_nccl_gpu.py`` (one rank per GPU, backend nccl = RCCL) or by hand on one GPU:

    SSI_LOCAL_DEVICE=0 SSI_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
        tests/workers/dp_step_worker.py --out gpurun_out/dp2.json

Rank 0 writes a JSON verdict; exit code 0 = the N-rank step equals the single-process step within bf16 rounding on every rank."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "speech-integration_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PARAMS = dict(vocab_size=700, num_layers=3, num_heads=4, num_kv_heads=2, embed_dim=256, max_seq_len=512, intermediate_dim=512)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    from oracle import hf_crosscheck as hx
    from ssi.distributed import GradSync, all_reduce_scalars, init_distributed
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    from ssi.optimizer import HipAdamW, scale_grads

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("SSI_LOCAL_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    init_distributed(device)
    sd = hx.seeded_state_dict(PARAMS, 77)
    loss_fn = CEWithChunkedOutputLoss()

    def build():
        m = HipLlamaDecoder(**PARAMS, dtype=torch.bfloat16, device=device)
        m.load_state_dict(sd)
        m.train()
        return m, HipAdamW(m.parameters(), model=m, lr=1e-2)

    def micro_batch(step, r):  # the micro-batch rank r sees at `step`
        return {k: v.to(device) for k, v in hx.seeded_batch(700, 2, 128, 1000 * step + r).items()}

    def fwd_bwd(model, batch, sync):
        n = (batch["labels"] != -100).sum()
        model.sync_this_backward = sync
        lb = compute_loss(batch, model, loss_fn) * n
        lb.backward()
        return int(n), float(lb)

    # ---- N ranks ----------------------------------------------------------------------------------------------------
    model, opt = build()
    gs = GradSync(model._flat_grad, model.buckets)
    model.grad_sync = gs
    losses = []
    grads_dp = None
    for step in range(args.steps):
        n, lb = fwd_bwd(model, micro_batch(step, rank), True)
        n_all, lb_all = all_reduce_scalars([n, lb], device, group=gs.scalar_group)
        gs.finish(defer_last=True)
        assert gs.deferred_range() is not None
        if step == 0:  # the exchanged gradients themselves (the deferred bucket drained first), before AdamW's sign-like update blurs them
            gs.finish_deferred()
            grads_dp = model._flat_grad.float().clone()
        scale_grads(model, torch.tensor(1.0 / n_all))
        opt.step()
        opt.zero_grad(set_to_none=True)
        losses.append(lb_all / n_all)
    torch.cuda.synchronize()
    flat = model._flat.float().clone()
    # every rank ends with the same weights, bit for bit
    other = flat.clone()
    dist.broadcast(other, src=0)
    same_across_ranks = bool(torch.equal(other, flat))

    # ---- one process, the union of the micro-batches in one accumulation window -----------------------------------------
    ref, ropt = build()
    ref_losses = []
    grad_rel = float("nan")
    for step in range(args.steps):
        n_all, lb_all = 0, 0.0
        for r in range(world):
            n, lb = fwd_bwd(ref, micro_batch(step, r), False)
            n_all, lb_all = n_all + n, lb_all + lb
        if step == 0:  # sum over the ranks' micro-batches: bf16 all-reduce of bf16 gradients vs bf16 accumulation in one buffer
            g_ref = ref._flat_grad.float()
            grad_rel = float((grads_dp - g_ref).norm() / g_ref.norm())
        scale_grads(ref, torch.tensor(1.0 / n_all))
        ropt.step()
        ropt.zero_grad(set_to_none=True)
        ref_losses.append(lb_all / n_all)
    torch.cuda.synchronize()
    rflat = ref._flat.float()
    moved = float(rflat.abs().max())  # scale of the weights
    diff = float((flat - rflat).abs().max())
    rel = float((flat - rflat).norm() / rflat.norm())
    loss_err = max(abs(a - b) / abs(b) for a, b in zip(losses, ref_losses))
    # AdamW moves every weight by about lr per step whatever the gradient's size, so a near-zero gradient whose sign differs between the
    # two summation orders (bf16 all-reduce of per-rank sums vs one accumulation window) shows up as a 2 x lr difference on a few
    # elements: bound the worst element by that, the bulk by the relative Frobenius error
    lr, far = 1e-2, float(((flat - rflat).abs() > 2e-3).float().mean())
    # more ranks = more bf16 roundings on both sides of the comparison = more flipped signs (weights: 2.6e-3 at 2 ranks, 7.3e-3 at 4, measured):
    # the weight bounds are therefore loose; what must hold at any world size is that the summed gradients agree to bf16 rounding, that every
    # rank ends on the same bits, and that no element moved further than AdamW can move it
    ok = (same_across_ranks and diff <= 2.5 * lr * args.steps and rel <= 5e-2 and far <= 0.1 and loss_err <= 5e-3 and grad_rel <= 2e-2)
    flags = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    verdict = {"backend": dist.get_backend(), "world": world, "same_across_ranks": same_across_ranks, "weights_rel_err": rel,
               "weights_max_abs_err": diff, "summed_gradient_rel_err": grad_rel, "weights_absmax": moved, "fraction_beyond_2e-3": far, "loss_rel_err": loss_err, "losses": losses, "ref_losses": ref_losses,
               "bytes_reduced": gs.bytes_reduced, "ok_all_ranks": bool(flags.item() == 1.0)}
    if rank == 0:
        print(json.dumps(verdict), flush=True)
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            json.dump(verdict, open(args.out, "w"))
    dist.barrier()
    dist.destroy_process_group()
    return 0 if verdict["ok_all_ranks"] else 1


if __name__ == "__main__":
    sys.exit(main())
