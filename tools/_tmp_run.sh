for k in 1 5 13; do echo "key8=$k"; DW_KEY8=$k GRAPHS=0 python tools/decode_profile.py | tail -1; done
