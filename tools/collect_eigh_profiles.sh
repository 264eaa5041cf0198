R=r04
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_$R; rm -rf $OUT; mkdir -p $OUT
E="python3 tools/eigh_r50.py"
EO=$OUT/eigh; mkdir -p $EO
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $EO/trace -- $E > $EO/trace.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $EO/fetch -- $E > $EO/fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $EO/write -- $E > $EO/write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $EO/sq -- $E > $EO/sq.log 2>&1
cp $(find $EO/trace -name "*kernel_stats.csv" | head -1) gpurun_out/${R}_eigh_kernel_stats.csv
python3 tools/parse_pmc.py $EO jacobi_two_sided32_kernel jacobi_cols_pair32_kernel eigh_norms32_kernel gemm_f64_macro_kernel jacobi_colsv_pair_kernel jacobi_rows_kernel jacobi_cols_kernel > gpurun_out/${R}_eigh_pmc.json
grep "factors in\|worst" $EO/trace.log > gpurun_out/${R}_eigh_run.txt
python3 tools/eigh_r50.py 2>&1 | grep "factors in\|worst" >> gpurun_out/${R}_eigh_run.txt
rm -rf $OUT
cat gpurun_out/${R}_eigh_run.txt
