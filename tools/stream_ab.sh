# same-box A/B of the single-image pass (final sources): round-6 path | small_head=0 | small_head=0 small_tail=0 (= round 5), then the kernel trace of one pass
cd $GRAFT_REPO_ROOT
export SEGVLAD_GUARD=0
for rep in 1 2; do
for opt in "" "small_head=0" "small_head=0 small_tail=0"; do
  echo "[$rep] options: ${opt:-(default)}"; timeout 120 python tools/probe_stream.py 50 $opt 2>/dev/null | tail -1
done
done
bash tools/trace_pass.sh 2>/dev/null | tail -7
