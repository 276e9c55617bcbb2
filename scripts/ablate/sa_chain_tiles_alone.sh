for t in off 1 4 10 1000; do
  if [ $t = off ]; then export REGNET_SA_OFF=1; unset REGNET_SA_CHAIN_TILES; else unset REGNET_SA_OFF; export REGNET_SA_CHAIN_TILES=$t; fi
  echo "== tiles $t"; REPS=30 ROWS=4 python scripts/features_alone.py 8 2>&1 | grep "sa_chain3\|feature stage"
done
