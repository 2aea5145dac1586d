"""Worker of tests/test_scripts_gpu.py::test_two_rank_train_main: one rank of `train.main` under torchrun
(collectives over gloo so that two ranks can share the one GPU of the test box)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    data_dir, out_dir = sys.argv[1], sys.argv[2]
    from ick_amd import train as tr
    rank = int(os.environ["RANK"])
    torch.manual_seed(1234 + rank)      # different random initial weights per rank: the broadcast must fix that
    seen = []
    def spy(batch):
        seen.append(batch[1].clone())   # the captions identify the samples

    tr._batch_hook = spy                # the TRAIN batches the prefetcher hands to the fused step
    steps = []

    class Recording(tr.TrainStep):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            steps.append(self)

    tr.TrainStep = Recording
    mods = []
    orig_bc = tr.dp.broadcast_module_state

    def spy_bc(modules, *a, **k):
        mods.extend(modules)
        return orig_bc(modules, *a, **k)

    tr.dp.broadcast_module_state = spy_bc
    cfg = tr.Config(variant="knowledge", data_dir=data_dir, data_name="toy", epochs=2, batch_size=4, workers=0,
                    print_freq=1000, fused=True, out_dir=out_dir if rank == 0 else os.path.join(out_dir, "r%d" % rank),
                    seed=5)
    os.makedirs(cfg.out_dir, exist_ok=True)
    hist = tr.main(cfg)
    import ick_amd.utils as ut
    torch.cuda.synchronize()
    # every rank reports its final weights (through rank 0's checkpoint for rank 0, directly for the others)
    torch.save({"hist": hist, "seen": seen}, os.path.join(out_dir, "rank%d.pt" % rank))
    import torch.distributed as dist
    torch.save(steps[-1].flat_p.detach().cpu().clone(), os.path.join(out_dir, "flat%d.pt" % rank))
    # the frozen Encoder.conv1 is outside the bucket: it must have been broadcast too
    torch.save(mods[0].conv1.weight.detach().cpu().clone(), os.path.join(out_dir, "conv%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()
