set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r6
echo "=== VAE parity (fused statistics on)"
timeout 900 python -m pytest tests -x -q -m gpu -k "vae or autoencoder or decode or encode or stable_diffusion_pipeline or animation_pipeline_call" 2>&1 | tail -4
echo "=== VAE decode timing: fused statistics off / on (tools/vae_only.py)"
for v in 0 1 0 1; do echo "-- FYC_VAE_FUSE_STATS=$v"; FYC_VAE_FUSE_STATS=$v timeout 300 python tools/vae_only.py 2>&1 | grep -v amdgpu.ids | tail -4; done
