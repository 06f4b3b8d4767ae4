cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
M2T_CONV_PIPE=0 python scratch/conv_variants.py plain 0
M2T_CONV_PIPE=0 python scratch/conv_variants.py persistent 1
M2T_CONV_PIPE=1 M2T_CONV_PIPE_BLOCKS=256 M2T_CONV_XCD=0 python scratch/conv_variants.py pipe256 0
M2T_CONV_PIPE=1 M2T_CONV_PIPE_BLOCKS=256 M2T_CONV_XCD=1 python scratch/conv_variants.py pipe256x 0
M2T_CONV_PIPE=1 M2T_CONV_PIPE_BLOCKS=512 M2T_CONV_XCD=1 python scratch/conv_variants.py pipe512x 0
python - <<'PY'
import torch
a = torch.load("gpurun_out/sr_plain.pt")
for t in ("persistent", "pipe256", "pipe256x", "pipe512x"):
    b = torch.load(f"gpurun_out/sr_{t}.pt")
    print(t, "equal" if torch.equal(a, b) else f"max abs diff {float((a-b).abs().max()):.3e}")
PY
rm -f gpurun_out/sr_*.pt
