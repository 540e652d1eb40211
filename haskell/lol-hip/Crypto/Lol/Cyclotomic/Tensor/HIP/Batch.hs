{-|
Module      : Crypto.Lol.Cyclotomic.Tensor.HIP.Batch
Description : Whole lists of ring elements as one device slab.

UNCOMPILED SOURCE (no GHC in the build image; see ../../../../../../README.md).

The 'Tensor' class moves one ring element per call: through 'HT' that is one PCIe round trip per
operation (INTEGRATION.md gives the measured cost).  The throughput is in the batch dimension —
the B leading polynomials every entry point of include/lolhip.h takes.  This module is the
Haskell face of that: a list of tensors becomes one [B][n][T] slab, one call, one slab back.
lol-apps' homomorphic operations over many ciphertexts (SymmSHE's (*), keySwitchQuadCirc,
rescaleCyc, tunnel) map to the pipeline entry points the same way (INTEGRATION.md section B).
-}

{-# LANGUAGE DataKinds           #-}
{-# LANGUAGE FlexibleContexts    #-}
{-# LANGUAGE PolyKinds           #-}
{-# LANGUAGE ScopedTypeVariables #-}

module Crypto.Lol.Cyclotomic.Tensor.HIP.Batch
( crtBatch, crtInvBatch, lBatch, lInvBatch, mulGPowBatch, mulGDecBatch, mulCRT, ringProduct
) where

import Data.Int
import qualified Data.Vector.Storable as SV

import Crypto.Lol.Cyclotomic.Tensor.HIP.Backend

-- | b vectors of n*T residues each -> one slab -> op -> b vectors
batched :: (SV.Storable r) => Plan -> Op -> [SV.Vector r] -> [SV.Vector r]
batched _ _ [] = []
batched plan op xs =
  let len  = SV.length (head xs)
      slab = opHost plan op (fromIntegral $ length xs) (SV.concat xs)
  in [ SV.slice (i * len) len slab | i <- [0 .. length xs - 1] ]

crtBatch, crtInvBatch, lBatch, lInvBatch, mulGPowBatch, mulGDecBatch
  :: (SV.Storable r) => Plan -> [SV.Vector r] -> [SV.Vector r]
crtBatch     p = batched p OpCRT
crtInvBatch  p = batched p OpCRTInv
lBatch       p = batched p OpL
lInvBatch    p = batched p OpLInv
mulGPowBatch p = batched p OpMulGPow
mulGDecBatch p = batched p OpMulGDec

-- | zipWith (*) over CRT-basis slabs (mulRq, mul.cpp:27-30; CPP.hs:257-260)
mulCRT :: (SV.Storable r) => Plan -> [SV.Vector r] -> [SV.Vector r] -> [SV.Vector r]
mulCRT plan as bs =
  let len  = SV.length (head as)
      slab = op2Host plan OpMul (fromIntegral $ length as) (SV.concat as) (SV.concat bs)
  in [ SV.slice (i * len) len slab | i <- [0 .. length as - 1] ]

-- | crtInv (crt a * crt b) per pair, powerful basis in and out: the fused kernel (one launch,
-- three slab passes) that bench.py measures — Cyc's (*) for two Pow-basis operands (Cyc.hs:262-297).
ringProduct :: (SV.Storable r) => Plan -> [SV.Vector r] -> [SV.Vector r] -> [SV.Vector r]
ringProduct plan as bs =
  let len  = SV.length (head as)
      slab = op2Host plan OpPolyMul (fromIntegral $ length as) (SV.concat as) (SV.concat bs)
  in [ SV.slice (i * len) len slab | i <- [0 .. length as - 1] ]
