cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scratch/conv_variants.py plain 0
python scratch/conv_variants.py persistent 1
python - <<'PY'
import torch
a = torch.load("gpurun_out/sr_plain.pt"); b = torch.load("gpurun_out/sr_persistent.pt")
print("persistent", "equal" if torch.equal(a, b) else f"max abs diff {float((a-b).abs().max()):.3e}")
PY
rm -f gpurun_out/sr_*.pt
