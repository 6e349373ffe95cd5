cd "$GRAFT_REPO_ROOT"
L=funscript_flow_amd/csrc/libffl_hip.so
cp $L /tmp/new.so
for r in 1 2; do
cp profiles/tools/_ab/libffl_ref.so $L && TOP=14 bash profiles/tools/exp.sh "ref$r|" || exit 1
cp /tmp/new.so $L && TOP=14 bash profiles/tools/exp.sh "new$r|" || exit 1
done
