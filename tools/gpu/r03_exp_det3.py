"""Uninitialised-memory hunt: fill the caching allocator's free blocks with NaN (allocate, fill, free), then tune the fused OPT block.
Anything that reads a torch.empty buffer before writing it turns the loss trace into NaN (or changes it)."""
import copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import transformers
from auto_round_amd.autoround import loss_mask_ids
from auto_round_amd.quantizer import BlockContext, SignRoundConfig, SignRoundQuantizer
from auto_round_amd.schemes import apply_scheme, resolve_scheme
from auto_round_amd.testing import t3_fixture as fx

dev = torch.device("cuda:0")


def poison(val):
    bufs = []
    for mb in (1, 2, 3, 6, 12, 24, 25, 48, 50, 75, 96, 100, 150, 200, 300, 400, 600, 800, 1200):
        for _ in range(3):
            bufs.append(torch.full((mb * 262144,), val, dtype=torch.float32, device=dev))
    for kb in (1, 4, 16, 64, 256, 512):
        for _ in range(8):
            bufs.append(torch.full((kb * 256,), val, dtype=torch.float32, device=dev))
    del bufs
    torch.cuda.synchronize()


model = fx.build_model("opt125m").to(dev)
for p in model.parameters():
    p.requires_grad_(False)
tokens = fx.calib_tokens("opt125m", 128, 2048)
block = fx.decoder_blocks(model)[0]
apply_scheme(block, resolve_scheme("W4A16"))
x0, others = fx.capture_block_inputs(model, block, tokens, dev)
ids = loss_mask_ids(tokens, None)
y = SignRoundQuantizer(SignRoundConfig(iters=1, batch_size=8, bits=4, fused_block=False), device=dev).calibrate_block(block, x0, others)
xs, ys = fx.sha(x0), fx.sha(y)
res = {}
for name, kw in dict(fused_graph={}, fused_no_graph=dict(hip_graph=False), module=dict(fused_block=False, mfma_dw_gemm=False)).items():
    traces = []
    for fill in (None, 0.0, float("nan"), 1e30, float("nan")):
        if fill is not None:
            poison(fill)
        blk = copy.deepcopy(block)
        cfg = dict(iters=12, batch_size=8, bits=4, fused_block=True, mfma_dw_gemm=True)
        cfg.update(kw)
        qz = SignRoundQuantizer(SignRoundConfig(**cfg), device=dev)
        transformers.set_seed(42)
        qz.quantize_block(blk, x0, others, y, None, BlockContext(0, 1, "0"), input_ids=ids)
        torch.cuda.synchronize()
        traces.append([float(v) for v in qz.last_stats["loss_trace"]])
        assert fx.sha(x0) == xs and fx.sha(y) == ys, "inputs or targets were modified"
    res[name] = traces
    for t in traces:
        print(name, ["%.9e" % v for v in t[:5]], flush=True)
json.dump(res, open(os.path.join(os.environ.get("OUT", "."), "det_poison.json"), "w"), indent=1)
