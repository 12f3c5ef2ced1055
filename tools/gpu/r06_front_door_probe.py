import os, sys, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from transformers import LlamaConfig, LlamaForCausalLM
from auto_round_amd.autoround import AutoRound
torch.manual_seed(0)
cfg = LlamaConfig(hidden_size=1024, intermediate_size=2048, num_attention_heads=8, num_key_value_heads=2, num_hidden_layers=2, vocab_size=512,
                  max_position_embeddings=4096, tie_word_embeddings=False)
cfg._attn_implementation = "sdpa"
model = LlamaForCausalLM(cfg).to(torch.bfloat16)
tokens = torch.randint(0, 512, (16, 2048), generator=torch.Generator().manual_seed(1))
ar = AutoRound(model, None, scheme="W4A16", iters=10, nsamples=16, seqlen=2048, batch_size=8, dataset=tokens)
t0 = time.time(); ar.quantize(); torch.cuda.synchronize()
rep = ar.quantizer.last_exact_report
print("exact:", ar.quantizer.last_exact, "plan:", {k: v for k, v in (rep or {}).get("plan", {}).items() if v}, "dropped:", (rep or {}).get("dropped"), "s:", round(time.time() - t0, 1))
