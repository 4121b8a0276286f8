# auromat-convert on a folder of 32 full-size JPEG frames (copies of the reference's test frame): frames per second with
# and without read-ahead decoding
D=/tmp/convert_probe; rm -rf $D; mkdir -p $D/in
R=$GRAFT_REPO_ROOT/tests/golden/resources
for i in $(seq -w 1 32); do cp $R/ISS030-E-102170_dc.jpg $D/in/f$i.jpg; cp $R/ISS030-E-102170_dc.wcs $D/in/f$i.wcs; done
cd $GRAFT_REPO_ROOT
for ahead in 1 8 1 8; do
  rm -rf $D/out
  s=$(date +%s.%N)
  AMT_CONVERT_READ_AHEAD=$ahead python -m auromat_amd.cli.convert --data $D/in --format netcdf --resample --grid geo --px-per-deg 10 --out $D/out --without-mag > /dev/null 2>&1
  e=$(date +%s.%N)
  echo "read-ahead $ahead: $(ls $D/out | wc -l) files in $(python3 -c "print(round($e-$s,2))") s (process start-up included)"
done
