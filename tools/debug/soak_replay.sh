# replay the soak failures of profiles/r15_soak_21000.txt on the current build
cd $GRAFT_REPO_ROOT
for s in 21028002 21093005 21109003; do python tools/debug/soak_classic.py $s 2>&1 | grep -v amdgpu | tail -2; done
for b in 21012 21022 21037 21054 21074 21091 21098 21105 21108 21116 21051 21056; do
  OMX_SOAK_SEED=$b python -m pytest tests/test_gpu_soak.py -q -m gpu 2>&1 | grep -E "passed|failed|AssertionError" | cut -c1-260 | sed "s/^/base $b: /"
done
