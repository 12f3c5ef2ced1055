"""Worker of tests/test_gpu_multi_rank.py: the standalone front door under a launcher, one process per rank
(`python -m torch.distributed.run --nproc-per-node N tests/multi_rank_front_door.py <out_dir> <device_map> <quanted_input 0|1> [scheme]`).
Every rank builds the same seeded tiny model and calibration tokens and calls `AutoRound(...).quantize_and_save(out_dir)`; rank 0 also
writes `<out_dir>/run.json` (who took part, what each rank tuned)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir, device_map, quanted = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
    scheme = sys.argv[4] if len(sys.argv) > 4 else "W4A16"
    from test_gpu_autoround import tiny_llama

    from auto_round_amd.autoround import AutoRound

    model = tiny_llama(layers=4)
    tokens = torch.randint(0, 512, (16, 64), generator=torch.Generator().manual_seed(1))
    kw = dict(group_size=32) if scheme == "W4A16" else {}
    ar = AutoRound(model, None, scheme=scheme, iters=12, nsamples=16, seqlen=64, batch_size=4, dataset=tokens,
                   enable_quanted_input=quanted, device_map=device_map, **kw)
    qmodel, _ = ar.quantize_and_save(out_dir)
    import torch.distributed as dist

    me = {"rank": ar.rank, "world": ar.world, "device": str(ar.device), "sharded": ar.sharded, "data_parallel": ar.data_parallel,
          "owned_blocks": list(ar.owned_blocks), "backend": dist.get_backend() if dist.is_initialized() else None,
          "losses": {str(k): ar.records[k]["stats"]["best_loss"] for k in ar.owned_blocks},
          # every rank ends with the WHOLE tuned model: a checksum of all tuned weights must agree across ranks
          "tuned_weights_checksum": float(sum(m.weight.double().abs().sum().item() for n, m in qmodel.named_modules()
                                              if isinstance(m, torch.nn.Linear) and hasattr(m, "scale")))}
    if dist.is_initialized():
        allme = [None] * ar.world
        dist.all_gather_object(allme, me)
        if ar.rank == 0:
            with open(os.path.join(out_dir, "run.json"), "w") as f:
                json.dump(allme, f, indent=1)
        dist.barrier()
        dist.destroy_process_group()
    else:
        with open(os.path.join(out_dir, "run.json"), "w") as f:
            json.dump([me], f, indent=1)


if __name__ == "__main__":
    main()
