# column-pass ablations (C5) + parity soak of the EXACT-combine variants
bash tools/c5_groups.sh "64" hip cnost cstonly cstonlyz 2>&1 | tail -12 > gpurun_out/r4d_cols.txt
cat gpurun_out/r4d_cols.txt
L=$PWD/topdogspectrumanalyser_amd
for lib in hip ex1 ex3; do
  echo "== $lib" >> gpurun_out/r4d_soak.txt
  TDSA_HIP_LIB=$L/libtdsa_$lib.so timeout 900 python tools/parity_soak.py --cases 1200 --long 0 2>&1 | tail -6 >> gpurun_out/r4d_soak.txt
done
cat gpurun_out/r4d_soak.txt
