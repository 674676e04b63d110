/* abi_check.c -- include/sangnom_hip.h must be usable from plain C (C99, -pedantic) and every declared entry
 * point must link.  Runs only what needs no GPU: the ABI version and Create_SangNom2's argument checks. */
#include <stdio.h>
#include <string.h>

#include "sangnom_hip.h"

int main(void)
{
    /* take the address of every entry point so that a missing export fails at link time */
    typedef void (*fn_t)(void);
    const fn_t fns[] = {(fn_t)sn_abi_version, (fn_t)sn_validate, (fn_t)sn_create, (fn_t)sn_destroy,
                         (fn_t)sn_last_error, (fn_t)sn_process_host, (fn_t)sn_process_device,
                         (fn_t)sn_process_device_strided, (fn_t)sn_host_slots, (fn_t)sn_submit_host,
                         (fn_t)sn_collect_host, (fn_t)sn_turn_device, (fn_t)sn_synchronize,
                         (fn_t)sn_get_stream, (fn_t)sn_get_info, (fn_t)sn_debug_read_pool,
                         (fn_t)sn_debug_read_coupled_rows, (fn_t)sn_debug_set_bands, (fn_t)sn_aa_create, (fn_t)sn_aa_process_host,
                         (fn_t)sn_aa_last_error, (fn_t)sn_aa_destroy, (fn_t)sn_pin_host_buffer, (fn_t)sn_unpin_host_buffer,
                         (fn_t)sn_submit_host_to, (fn_t)sn_create_with_policy, (fn_t)sn_get_policy, (fn_t)sn_set_policy,
                         (fn_t)sn_aa_create_with_policy, (fn_t)sn_debug_raise_chain_fault};
    sn_config c;
    char msg[256];
    size_t i;
    for (i = 0; i < sizeof fns / sizeof fns[0]; ++i)
        if (!fns[i]) return 1;
    if (sn_abi_version() != SN_ABI_VERSION) return 2;
    memset(&c, 0, sizeof c);
    c.struct_size = (int32_t)sizeof c;
    c.width = 64;
    c.height = 32;
    c.bytes_per_sample = 1;
    c.bits_per_sample = 8;
    c.num_planes = 1;
    c.order = 1;
    c.aa = 48;
    c.luma = c.chroma = 1;
    if (sn_validate(&c, msg, sizeof msg) != SN_OK) return 3;
    c.aa = 200;
    if (sn_validate(&c, msg, sizeof msg) != SN_ERR_CONFIG || strcmp(msg, "SangNom2: aa must be between 0..128.") != 0) return 4;
    c.aa = 48;
    c.height = 31;
    if (sn_validate(&c, msg, sizeof msg) != SN_ERR_CONFIG || strcmp(msg, "SangNom2: height must be even.") != 0) return 5;
    printf("abi %d ok\n", sn_abi_version());
    return 0;
}
