cd /root/repo
export PYTHONPATH=/root/repo
timeout 600 python -m pytest tests/test_gpu_embed.py -x -q 2>&1 | tail -5
timeout 200 python tools/r03_frontend_ab.py 10 2>&1 | grep "embed" | sed 's/^/product: /'
for b in 1; do
FVTA_LIB_PATH=$PWD/fvta_memexqa_amd/csrc/diag/libfvta_hip_embed_abl$b.so timeout 200 python tools/r03_frontend_ab.py 10 2>&1 | grep "embed" | sed "s/^/abl $b: /"
done
