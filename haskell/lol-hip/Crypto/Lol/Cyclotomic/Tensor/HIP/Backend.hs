{-|
Module      : Crypto.Lol.Cyclotomic.Tensor.HIP.Backend
Description : FFI layer over liblolhip (include/lolhip.h) and the plan cache.

UNCOMPILED SOURCE (no GHC in the build image; see ../../../../../../README.md).

Counterpart of lol-cpp's @CPP/Backend.hs@ (reference lines 155-337): where that module binds one
C symbol per (operation, element type) and re-marshals moduli and twiddles on every call, this one
binds the plan API: a plan is created once per (prime powers, moduli) and owns what CT passes per
call.  Element marshalling ('ZqTuple', 'getModuli', 'numComponents', 'CPP') is lol-cpp's.
-}

{-# LANGUAGE FlexibleContexts         #-}
{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE ScopedTypeVariables      #-}

module Crypto.Lol.Cyclotomic.Tensor.HIP.Backend
( PlanH, ExtH, Plan, Ext, Op(..), ExtOp(..)
, planFor, extFor, planHasCRT
, opHost, opHostMaybe, op2Host, extHost
, ok, errNotDivisible
) where

import Control.Concurrent.MVar
import Control.Monad
import qualified Data.Map.Strict as M
import Data.Int
import qualified Data.Vector.Storable         as SV
import qualified Data.Vector.Storable.Mutable as SM
import Foreign.C.Types
import Foreign.ForeignPtr
import Foreign.Marshal.Alloc (alloca)
import Foreign.Marshal.Array (withArrayLen)
import Foreign.Ptr
import Foreign.Storable
import System.IO.Unsafe (unsafePerformIO)

import Crypto.Lol.Cyclotomic.Tensor.CPP.Backend (CPP, marshalFactors)

data PlanH      -- lolhip_plan
data ExtH       -- lolhip_ext
type Plan = ForeignPtr PlanH
type Ext  = ForeignPtr ExtH

ok, errNotDivisible :: CInt
ok = 0
errNotDivisible = -7          -- LOLHIP_ERR_NOT_DIVISIBLE: divG* would return 0 (CPP.hs:309-323)

-- include/lolhip.h: LOLHIP_OP_*
data Op = OpCRT | OpCRTInv | OpMul | OpPolyMul | OpL | OpLInv | OpMulGPow | OpMulGDec
        | OpDivGPow | OpDivGDec | OpMulGCRT | OpDivGCRT deriving (Enum, Eq, Show)
-- include/lolhip.h: LOLHIP_EXT_*
data ExtOp = ExtTwacePowDec | ExtTwaceCRT | ExtEmbedPow | ExtEmbedDec | ExtEmbedCRT | ExtCoeffs
           deriving (Enum, Eq, Show)

foreign import ccall unsafe "lolhip_plan_create"
  c_planCreate :: Ptr CPP -> CInt -> Ptr Int64 -> CInt -> CInt -> Ptr (Ptr PlanH) -> IO CInt
foreign import ccall unsafe "&lolhip_plan_destroy"
  p_planDestroy :: FunPtr (Ptr PlanH -> IO ())
foreign import ccall unsafe "lolhip_plan_has_crt"
  c_planHasCRT :: Ptr PlanH -> IO CInt
foreign import ccall unsafe "lolhip_ext_create"
  c_extCreate :: Ptr PlanH -> Ptr PlanH -> Ptr (Ptr ExtH) -> IO CInt
foreign import ccall unsafe "&lolhip_ext_destroy"
  p_extDestroy :: FunPtr (Ptr ExtH -> IO ())
-- host-pointer entry points: one H2D, the kernels, one D2H on a leased stream.  They BLOCK in
-- hipStreamSynchronize for the whole kernel time, so they are `safe` imports: an `unsafe` call would hold the
-- capability (no other Haskell thread runs on it, GC waits) until the GPU is done.  Safe calls may migrate
-- between OS threads; liblolhip's staging sets belong to no thread (include/lolhip.h, lolhip_thread_release).
foreign import ccall safe "lolhip_op_host"
  c_opHost :: Ptr PlanH -> CInt -> Ptr Int64 -> Ptr Int64 -> Int64 -> IO CInt
foreign import ccall safe "lolhip_ext_host"
  c_extHost :: Ptr ExtH -> CInt -> Ptr Int64 -> Ptr Int64 -> Int64 -> IO CInt

-- | One plan per (prime powers, moduli), created on first use and kept for the life of the
-- process: where CT recomputes @ru@/@ruInv@ per type (CPP.hs:422-442) and re-marshals the moduli
-- per call (Backend.hs:195-199), the plan holds them in HBM.
{-# NOINLINE planCache #-}
planCache :: MVar (M.Map ([(Int16, Int16)], [Int64]) Plan)
planCache = unsafePerformIO $ newMVar M.empty

planFor :: [(Int, Int)] -> [Int64] -> Plan
planFor pps qs = unsafePerformIO $ modifyMVar planCache $ \m ->
  let key = (map (\(p, e) -> (fromIntegral p, fromIntegral e)) pps, qs)
  in case M.lookup key m of
       Just p  -> return (m, p)
       Nothing -> do
         p <- SV.unsafeWith (marshalFactors pps) $ \pfac ->
              withArrayLen qs $ \t pq ->
              alloca $ \out -> do
                rc <- c_planCreate pfac (fromIntegral $ length pps) pq (fromIntegral t) 0 out
                when (rc /= ok) $ error $ "lolhip_plan_create: status " ++ show rc
                peek out >>= newForeignPtr p_planDestroy
         return (M.insert key p m, p)

planHasCRT :: Plan -> Bool
planHasCRT p = unsafePerformIO $ withForeignPtr p $ fmap (/= 0) . c_planHasCRT

-- | Tables for the ring extension m | m'; both plans must share their moduli.  Cached like the plans (which
-- live as long as the process, so their addresses identify them): building one costs several hipMalloc and
-- blocking copies of index tables, and twace/embed/coeffs ask for it on EVERY element operation.
{-# NOINLINE extCache #-}
extCache :: MVar (M.Map (Ptr PlanH, Ptr PlanH) Ext)
extCache = unsafePerformIO $ newMVar M.empty

extFor :: Plan -> Plan -> Ext
extFor lo hi = unsafePerformIO $ modifyMVar extCache $ \m ->
  withForeignPtr lo $ \plo -> withForeignPtr hi $ \phi ->
    case M.lookup (plo, phi) m of
      Just e  -> return (m, e)
      Nothing -> alloca $ \out -> do
        rc <- c_extCreate plo phi out
        when (rc /= ok) $ error $ "lolhip_ext_create: status " ++ show rc
        e <- peek out >>= newForeignPtr p_extDestroy
        return (M.insert (plo, phi) e m, e)

-- | In-place operation on a thawed copy, as CT does (CPP.hs:325-337): Haskell owns the memory,
-- the library never keeps the pointer.  @b@ polynomials of @n*T@ Int64 residues each.
opHostMaybe :: (Storable r) => Plan -> Op -> Int64 -> SV.Vector r -> Maybe (SV.Vector r)
opHostMaybe plan op b x = unsafePerformIO $ do
  y <- SV.thaw x
  rc <- withForeignPtr plan $ \pp -> SM.unsafeWith y $ \py ->
          c_opHost pp (fromIntegral $ fromEnum op) (castPtr py) nullPtr b
  if rc == ok then Just <$> SV.unsafeFreeze y
  else if rc == errNotDivisible then return Nothing
  else error $ "lolhip_op_host " ++ show op ++ ": status " ++ show rc

opHost :: (Storable r) => Plan -> Op -> Int64 -> SV.Vector r -> SV.Vector r
opHost plan op b = maybe (error "lolhip: unexpected NOT_DIVISIBLE") id . opHostMaybe plan op b

-- | Two-operand form (OpMul: zipWithT (*); OpPolyMul: crtInv (crt a * crt b))
op2Host :: (Storable r) => Plan -> Op -> Int64 -> SV.Vector r -> SV.Vector r -> SV.Vector r
op2Host plan op b x z = unsafePerformIO $ do
  y <- SV.thaw x
  rc <- withForeignPtr plan $ \pp -> SM.unsafeWith y $ \py -> SV.unsafeWith z $ \pz ->
          c_opHost pp (fromIntegral $ fromEnum op) (castPtr py) (castPtr pz) b
  when (rc /= ok) $ error $ "lolhip_op_host " ++ show op ++ ": status " ++ show rc
  SV.unsafeFreeze y

-- | Out-of-place gather between O_m and O_m' (twace*/embed*/coeffs); @outLen@ in elements of r.
extHost :: (Storable r) => Ext -> ExtOp -> Int64 -> Int -> SV.Vector r -> SV.Vector r
extHost ext op b outLen x = unsafePerformIO $ do
  y <- SM.new outLen
  rc <- withForeignPtr ext $ \pe -> SM.unsafeWith y $ \py -> SV.unsafeWith x $ \px ->
          c_extHost pe (fromIntegral $ fromEnum op) (castPtr py) (castPtr px) b
  when (rc /= ok) $ error $ "lolhip_ext_host " ++ show op ++ ": status " ++ show rc
  SV.unsafeFreeze y
