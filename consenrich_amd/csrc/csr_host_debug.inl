// csr_host_debug.inl -- part of csr_lib.hip (one translation unit; included in this order): debugging aids (not part of the public ABI)
// clang-format off is NOT needed; this file is plain C++/HIP host code.

// ---------------------------------------------------------------------------------------------------------------
// debugging aids (not part of the public ABI)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int csr_debug_chain_step(csr_ctx *c, int kind, int op, int which, uint32_t flags, int force, unsigned int *count) {
    CHECK(need(c));
    Prm p = c->p;
    p.flags = flags;
    p.debugForce = force;
    p.warm = kind == 0 ? c->warmP : (kind == 1 ? c->warmX : c->warmB);
    const int grid = (int)c->NG;
    CHECK(settle(c));
    p.rerunCount = reinterpret_cast<unsigned int *>(c->dMail) + ST_DEBUG;
    if (op == 0) {
        if (kind == 0) hipLaunchKernelGGL(k_chain_spec<FwdPTrend>, dim3(grid), dim3(64), 0, c->stream, p);
        if (kind == 1) hipLaunchKernelGGL(k_chain_spec<FwdXTrend>, dim3(grid), dim3(64), 0, c->stream, p);
        if (kind == 2) hipLaunchKernelGGL(k_chain_spec<BwdTrend>, dim3(grid), dim3(64), 0, c->stream, p);
    } else {
        if (kind == 0) hipLaunchKernelGGL(k_chain_fix<FwdPTrend>, dim3(grid), dim3(64), 0, c->stream, p, which);
        if (kind == 1) hipLaunchKernelGGL(k_chain_fix<FwdXTrend>, dim3(grid), dim3(64), 0, c->stream, p, which);
        if (kind == 2) hipLaunchKernelGGL(k_chain_fix<BwdTrend>, dim3(grid), dim3(64), 0, c->stream, p, which);
    }
    LAUNCH_CHECK("debug chain step");
    CHECK(read_mail(c, MAIL_HDR));
    const unsigned int fresh = take_fresh(c, ST_DEBUG);
    if (count) *count = fresh;
    c->haveFwd = true;
    return 0;
}
extern "C" int csr_debug_read(csr_ctx *c, int buf, void *dst, int64_t bytes) {
    CHECK(need(c));
    const void *src = nullptr;
    switch (buf) {
        case 0: src = c->p.carryIn; break;
        case 1: src = c->p.carryOutA; break;
        case 2: src = c->p.carryOutB; break;
        case 3: src = c->p.tPf; break;
        case 4: src = c->p.tSZ; break;          // (S0u, zbar) records
        default: return fail("bad debug buffer");
    }
    HIPOK(hipMemcpy(dst, src, (size_t)bytes, hipMemcpyDeviceToHost));
    return 0;
}

