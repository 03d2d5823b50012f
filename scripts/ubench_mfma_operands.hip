// pure-asm MFMA loops: does the register class of the A/B operands change the issue rate?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","a0","a1","a2","a3","a4","a5","a6","a7"
#define M(d, A, B) "v_mfma_f32_32x32x2_f32 v[" d "], " A ", " B ", v[" d "]\n"
#define BODY(A0,B0,A1,B1,A2,B2,A3,B3) M("0:15",A0,B0) M("16:31",A1,B1) M("32:47",A2,B2) M("48:63",A3,B3) M("64:79",A0,B1) M("80:95",A1,B2) M("96:111",A2,B3) M("112:127",A3,B0)
template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, unsigned long long* cyc) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile(BODY("v128","v129","v128","v129","v128","v129","v128","v129") ::: CLOB);
        if (KIND == 1) asm volatile(BODY("v128","v129","v130","v131","v132","v133","v130","v129") ::: CLOB);
        if (KIND == 2) asm volatile(BODY("a0","v129","a1","v131","a2","v133","a3","v129") ::: CLOB);
        if (KIND == 3) asm volatile(BODY("v128","a4","v130","a5","v132","a6","v130","a7") ::: CLOB);
        if (KIND == 4) asm volatile(BODY("a0","a4","a1","a5","a2","a6","a3","a7") ::: CLOB);
        if (KIND == 5) asm volatile(BODY("v16","v32","v48","v64","v80","v96","v112","v0") ::: CLOB);  // A/B alias accumulators' banks
        if (KIND == 6) asm volatile(BODY("v129","v129","v133","v133","v129","v133","v133","v129") ::: CLOB);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char* n, unsigned long long* cyc) {
    k<KIND><<<256, 256>>>(10, cyc);
    k<KIND><<<256, 256>>>(4000, cyc);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s %.3f ticks per MFMA\n", n, (double)c / (4000 * 8.0));
}
int main() {
    unsigned long long* cyc; hipMalloc(&cyc, 8);
    run<0>("A,B fixed VGPRs", cyc); run<1>("A,B varying VGPRs", cyc); run<2>("A in AGPRs", cyc); run<3>("B in AGPRs", cyc);
    run<4>("A and B in AGPRs", cyc); run<5>("A,B = accumulator registers", cyc); run<6>("A == B register", cyc);
    return 0;
}
