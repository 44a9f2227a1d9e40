/* Compiled as strict C99 by tests/test_abi_and_host.py: include/consenrich_amd.h must be a plain-C header and the shared
 * library must link and answer from C (the drop-in boundary is a C ABI; no GPU is touched here). */
#include <stdio.h>
#include <string.h>

#include "consenrich_amd.h"

int main(void) {
    csr_model mdl;
    csr_ecm_cfg ecm;
    csr_bg_cfg bg;
    csr_bg_out bgo;
    csr_run_stats rs;
    memset(&mdl, 0, sizeof mdl);
    memset(&ecm, 0, sizeof ecm);
    memset(&bg, 0, sizeof bg);
    memset(&bgo, 0, sizeof bgo);
    memset(&rs, 0, sizeof rs);
    printf("abi %d devices %d sizes %u %u %u %u %u %u\n", csr_abi_version(), csr_device_count(), (unsigned)sizeof(csr_model),
           (unsigned)sizeof(csr_ecm_cfg), (unsigned)sizeof(csr_ecm_out), (unsigned)sizeof(csr_bg_cfg), (unsigned)sizeof(csr_bg_out),
           (unsigned)sizeof(csr_run_stats));
    /* argument validation happens before any device work */
    if (csr_solve_background(0, NULL, NULL, NULL, 1.0, 0.0, 0, 0, NULL, NULL, NULL) == 0) return 2;
    if (strlen(csr_last_error()) == 0) return 3;
    if (csr_format_bedgraph("chr1", 4, NULL, NULL, 0, 1, 0, NULL, 0, NULL, 0) >= 0) return 4;
    return 0;
}
