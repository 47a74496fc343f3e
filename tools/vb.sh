# usage: bash tools/vb.sh "<libs...>" "<env counts...>"   ("-" = the in-tree library)
for v in $1; do for n in $2; do if [ "$v" = "-" ]; then python tools/kbench.py $n 2>/dev/null | tail -1; else PARC_ENV_LIB=$v python tools/kbench.py $n 2>/dev/null | tail -1; fi; done; done
