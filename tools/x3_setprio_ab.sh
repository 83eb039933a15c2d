for r in 1 2 3; do for v in base setprio setprio16; do echo -n "$v "; timeout -k 10 60 tools/bin/x3s_$v 256000 | tail -1 || exit 1; done; done
