#!/usr/bin/env python3
"""Operand-rounding error of Winograd forms of one tower layer (3x3, 32 -> 32 channels) against the direct fp16-operand form
the tower runs.  All forms accumulate in fp32-or-better; what differs is WHICH values are rounded to fp16:
    direct        x, w rounded to fp16 (x is stored in fp16 anyway)
    F(2,3)  1-D   V = B^T d (along x) and U = G g rounded to fp16
    F(2x2,3x3)    V = B^T d B and U = G g G^T rounded to fp16
Reported: RMS error of the layer output against float64 on N(0,1) activations / weights of the tower's scale, relative to the
direct form's.   python scripts/winograd_error_probe.py"""
import numpy as np

rng = np.random.default_rng(0)
C, H, W = 32, 96, 128
x = rng.standard_normal((C, H, W)).astype(np.float16).astype(np.float64)          # the tensor holds fp16 values
w = (rng.standard_normal((C, C, 3, 3)) / 17.0)
q = lambda a: a.astype(np.float16).astype(np.float64)

def conv_direct(x, w):
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    out = np.zeros((C, H, W))
    for ky in range(3):
        for kx in range(3):
            out += np.einsum("oc,chw->ohw", w[:, :, ky, kx], xp[:, ky:ky + H, kx:kx + W])
    return out

exact = conv_direct(x, w)
direct = conv_direct(x, q(w))
e_direct = np.sqrt(np.mean((direct - exact) ** 2))

Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)

# 1-D along x: tiles of 2 output columns
xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
out1 = np.zeros((C, H, W))
U1 = q(np.einsum("ik,ocyk->ocyi", G, w))                                   # [o][c][ky][4]
for t in range(W // 2):
    d = xp[:, :, 2 * t:2 * t + 4]                                          # [c][H+2][4]
    V = q(np.einsum("ik,chk->chi", Bt, d))                                  # rounded transformed input
    M = np.zeros((C, H, 4))
    for ky in range(3):
        M += np.einsum("oci,chi->ohi", U1[:, :, ky, :], V[:, ky:ky + H, :])
    out1[:, :, 2 * t:2 * t + 2] = np.einsum("ji,ohi->ohj", At, M)
e_1d = np.sqrt(np.mean((out1 - exact) ** 2))

# 2-D
out2 = np.zeros((C, H, W))
U2 = q(np.einsum("ik,ockl,jl->ocij", G, w, G))
for ty in range(H // 2):
    for tx in range(W // 2):
        d = xp[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
        V = q(np.einsum("ik,ckl,jl->cij", Bt, d, Bt))
        M = np.einsum("ocij,cij->oij", U2, V)
        out2[:, 2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = np.einsum("ai,oij,bj->oab", At, M, At)
e_2d = np.sqrt(np.mean((out2 - exact) ** 2))
print(f"RMS output error vs float64 (output RMS {np.sqrt(np.mean(exact ** 2)):.3f}):")
print(f"  direct fp16 operands        {e_direct:.3e}   1.00x")
print(f"  Winograd F(2,3) along x     {e_1d:.3e}   {e_1d / e_direct:.2f}x   (24 instead of 36 MFMAs per 64 px)")
print(f"  Winograd F(2x2,3x3)         {e_2d:.3e}   {e_2d / e_direct:.2f}x   (16 instead of 36)")
