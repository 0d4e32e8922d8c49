R=$GRAFT_REPO_ROOT
cd /tmp
for ctx in 400 900; do
  A="--model 8b --steps 64 --warmup 8 --ctx $ctx --cap 1024"
  for rep in 1 2; do
    echo "ctx $ctx prev    : $(LD_LIBRARY_PATH=$R/tools/variants/prev timeout -k 10 120 $R/tools/step_bench $A | tail -1)"
    echo "ctx $ctx current : $(timeout -k 10 120 $R/tools/step_bench $A | tail -1)"
  done
done
