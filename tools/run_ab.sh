for rep in 1 2 3; do
  echo "##### process $rep"
  SK_STAGGER=0 ABLATE_ONLY="2 mates" ABLATE_ROUNDS=1 SK_LIBS=tools/ab/nt3.so python tools/ablate.py 8000000 2>&1 | grep -E "tile:|---" | grep -v "^--- cur" 
done
