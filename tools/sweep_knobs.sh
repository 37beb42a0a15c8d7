run() { echo -n "$1 "; env $1 timeout -k 10 100 python tools/shard_times.py c2 2>/dev/null | grep -E "G=[18]:" | sed -e 's/rank-0 share //' -e 's/ wall.*//' | paste - - ; }
for r in 1 2; do
run X=0
run HIPEXEC_LANES=1
run HIPEXEC_LANES=3
run HIPEXEC_FUSE_SHADE=0
run HIPEXEC_FUSE_SHADE=2
run HIPEXEC_UNIFORM_FETCH=0
run HIPEXEC_PACKET=0
run HIPEXEC_TAIL_MAX=0
run HIPEXEC_TAIL_MAX=600000
run HIPEXEC_OVERLAP=0
run HIPEXEC_OVERLAP=2
done
