#!/bin/bash
# round 3: counter passes over the front-end kernels alone (tools/r03_frontend_ab.py) and over the ragged variant's step
# (the weights-stationary backward step at its working point) -> gpurun_out/r03_pmc_fe/{frontend,ragged}.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH=$GRAFT_REPO_ROOT
out=gpurun_out/r03_pmc_fe
mkdir -p $out
pass() { tag=$1; name=$2; shift 2; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$tag/$name -o p -- $CMD > $out/$tag.$name.log 2> $out/$tag.$name.err; }
CMD="python3 tools/r03_frontend_ab.py 5"
pass fe sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass fe sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE
pass fe fetch FETCH_SIZE
pass fe write WRITE_SIZE
python3 tools/pmc_summary.py $out/frontend.json $out/fe/sq $out/fe/sq2 $out/fe/fetch $out/fe/write
CMD="python3 bench.py --variant ragged --steps 3 --warmup 1 --no-cpu-baseline"
pass rag sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass rag fetch FETCH_SIZE
pass rag write WRITE_SIZE
python3 tools/pmc_summary.py $out/ragged.json $out/rag/sq $out/rag/fetch $out/rag/write | grep -i "lstm"
