// Every compile-time hook that exists only to TIME a variant of a kernel (some of them knowingly UNSAFE: no cross-wave ordering,
// loosened or missing waits, results left unconverted) is named here and refused outside an ablation build: scripts/build_exp.sh passes
// -DDN_ABLATION_BUILD, the Makefile never does - so a stray -D cannot put one into the shipped library.  The hooks themselves sit next
// to the code they switch (mlp_device.h Pipe / Pipe48, mlp_stage48.h, mlp_fused48.hip, mlp_train.hip); what each measured is in
// HISTORY.md and profiles/.
//   DN_EXP_NODMA / NOREAD / NOBARRIER / SHALLOW / LOOSEWAIT / REGSTAGE / ROTATE / SETPRIO / NOPIN / NOSAVE / NOSETTLE   weight pipeline
//   DN_EXP_NOEPI / NOWAIT / NOTOP / HALF / ONLY_PAPER / TF_NOMASK / TF_NOUNIT, DN_G48_EPI_PIN / NO_XS / PREFETCH / COMPILER_READS / SYMMETRIC_DMA / PRIO /
//   BARRIER_EVERY_PHASE / SKIP_PE_FROM_LDS                                                                              48-point kernel
//   DN_STAMP, DN_STORE_POLICY_ID, DN_WG_*                                                                               stamps, training kernels
#pragma once
#if !defined(DN_ABLATION_BUILD) && \
    (defined(DN_EXP_NODMA) || defined(DN_EXP_NOREAD) || defined(DN_EXP_SHALLOW) || defined(DN_EXP_LOOSEWAIT) ||    \
     defined(DN_EXP_NOBARRIER) || defined(DN_EXP_REGSTAGE) || defined(DN_EXP_ROTATE) || defined(DN_EXP_SETPRIO) || \
     defined(DN_EXP_NOPIN) || defined(DN_EXP_NOSAVE) || defined(DN_EXP_NOSETTLE) || defined(DN_STAMP) || defined(DN_STORE_POLICY_ID) || \
     defined(DN_WG_NOREAD) || defined(DN_WG_NOSTAGE) || defined(DN_WG_STAMP) || defined(DN_WG_EPI) || defined(DN_WG_LOAD_POLICY_ID) ||  \
     defined(DN_WG_ONLY) || defined(DN_G48_PREFETCH) || defined(DN_G48_COMPILER_READS) || defined(DN_G48_SYMMETRIC_DMA) ||               \
     defined(DN_G48_PRIO) || defined(DN_G48_BARRIER_EVERY_PHASE) || defined(DN_G48_SKIP_PE_FROM_LDS) ||                                 \
     defined(DN_EXP_NOEPI) || defined(DN_EXP_NOWAIT) || defined(DN_EXP_NOTOP) || defined(DN_G48_EPI_PIN) || defined(DN_EXP_HALF) || defined(DN_EXP_ONLY_PAPER) || defined(DN_G48_NO_XS) || defined(DN_EXP_TF_NOMASK) || defined(DN_EXP_TF_NOUNIT) ||                                                          \
     (defined(DN_PREFETCH) && !defined(DN_PREFETCH_SET_BY_KERNEL_SOURCE)))
#error "DN_EXP_* / DN_WG_* / DN_G48_* / DN_PREFETCH / DN_STORE_POLICY_ID are ablation hooks: build them with scripts/build_exp.sh (-DDN_ABLATION_BUILD), never into libdexnerf_hip.so"
#endif
