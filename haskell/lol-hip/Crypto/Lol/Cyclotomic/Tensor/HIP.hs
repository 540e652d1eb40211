{-|
Module      : Crypto.Lol.Cyclotomic.Tensor.HIP
Description : 'Tensor' instance backed by liblolhip (hand-written HIP kernels for gfx950).

UNCOMPILED SOURCE (no GHC in the build image; see ../../../../../README.md).

@HT m r@ wraps lol-cpp's @CT m r@.  For element types that are Z_q or RNS tuples of Z_q the
linear maps of the class run on the GPU through a cached plan; everything else is CT's code.
Method by method this follows the reference instance (CPP.hs:204-264):

  l, lInv, mulGPow, mulGDec          basicDispatch dl ...      ->  lolhip_op_host OP_L ...
  divGPow, divGDec                   dispatchGInv              ->  the same; status -7 = Nothing
  crtFuncs (scalarCRT, mulGCRT, divGCRT, crt, crtInv)          ->  OP_MULGCRT, OP_DIVGCRT, OP_CRT, OP_CRTINV;
                                                                   Nothing iff lolhip_plan_has_crt = 0
  zipWithT                           SV.zipWith                ->  unchanged (the class cannot see that f = (*));
                                                                   Cyc's ring product goes through HIP.Batch.mulCRT
  twacePowDec, embedPow, embedDec, crtExtFuncs, coeffs         ->  lolhip_ext_host (index tables in HBM)
  scalarPow, powBasisPow, crtSetDec, tGaussianDec, gSqNormDec,
  fmapT, unzipT, entail*                                       ->  CT
-}

{-# LANGUAGE ConstraintKinds       #-}
{-# LANGUAGE DataKinds             #-}
{-# LANGUAGE FlexibleContexts      #-}
{-# LANGUAGE FlexibleInstances     #-}
{-# LANGUAGE InstanceSigs          #-}
{-# LANGUAGE MultiParamTypeClasses #-}
{-# LANGUAGE PolyKinds             #-}
{-# LANGUAGE ScopedTypeVariables   #-}
{-# LANGUAGE TypeFamilies          #-}
{-# LANGUAGE UndecidableInstances  #-}

module Crypto.Lol.Cyclotomic.Tensor.HIP (HT, HipElt(..)) where

import Control.DeepSeq
import Data.Constraint
import Data.Int
import qualified Data.Vector.Storable as SV

import Crypto.Lol.CRTrans
import Crypto.Lol.Cyclotomic.Tensor
import Crypto.Lol.Cyclotomic.Tensor.CPP            (CT, toVec, fromVec)   -- the two exports of ../../../../lol-cpp-exports.patch
import Crypto.Lol.Cyclotomic.Tensor.CPP.Backend    (Dispatch, ZqTuple, getModuli, numComponents)
import Crypto.Lol.Cyclotomic.Tensor.HIP.Backend
import Crypto.Lol.Prelude                          as LP
import Crypto.Lol.Types.Unsafe.ZqBasic             (ZqBasic)

-- | A tensor over the GPU backend.  Same representation as CT (a storable vector of @r@ in
-- the reference's layout: coefficient j of RNS component t at index j*T + t, tensor.h:69).
newtype HT (m :: Factored) r = HT (CT m r) deriving (Eq, Show, NFData)

-- | Does this element type take the GPU path, and with which moduli?  Z_q over Int64 and nested
-- pairs of such (lol-cpp's 'ZqTuple', Backend.hs:134-149); every other type answers Nothing and
-- stays on CT.
class HipElt r where
  hipModuli :: proxy r -> Maybe [Int64]
  hipModuli _ = Nothing

instance {-# OVERLAPPABLE #-} HipElt r
instance (Reflects q Int64) => HipElt (ZqBasic q Int64) where
  hipModuli _ = Just [proxy value (Proxy :: Proxy q)]
instance (HipElt a, HipElt b) => HipElt (a, b) where
  hipModuli _ = (++) <$> hipModuli (Proxy :: Proxy a) <*> hipModuli (Proxy :: Proxy b)

-- the plan of (m, r), or Nothing for element types that stay on the CPU
planOf :: forall m r . (Fact m, HipElt r) => Proxy '(m, r) -> Maybe Plan
planOf _ = planFor (proxy ppsFact (Proxy :: Proxy m)) <$> hipModuli (Proxy :: Proxy r)

-- run a plan operation on the vector inside a CT (CT's own conversions: toCT / SV.Vector r)
onGPU :: forall m r . (Fact m, HipElt r, TElt CT r)
      => Op -> (CT m r -> CT m r) -> HT m r -> HT m r
onGPU op cpu (HT x) = case planOf (Proxy :: Proxy '(m, r)) of
  Nothing   -> HT (cpu x)
  Just plan -> HT $ fromVec $ opHost plan op 1 (toVec x)

onGPUMaybe :: forall m r . (Fact m, HipElt r, TElt CT r)
           => Op -> (CT m r -> Maybe (CT m r)) -> HT m r -> Maybe (HT m r)
onGPUMaybe op cpu (HT x) = case planOf (Proxy :: Proxy '(m, r)) of
  Nothing   -> HT <$> cpu x
  Just plan -> HT . fromVec <$> opHostMaybe plan op 1 (toVec x)

-- CT <-> storable vector.  lol-cpp exports CT abstractly; these two are the only additions the
-- lol-cpp package needs for this backend (its CT' newtype unwrapped): `unCT . toCT'` and `CT . CT'`.
-- toVec / fromVec (CT m r <-> the Storable coefficient vector) come from lol-cpp once lol-cpp-exports.patch is applied:
-- the constructors of CT are private to CPP.hs (CPP.hs:33,86-96).

instance Tensor HT where

  type TElt HT r = (TElt CT r, HipElt r)

  entailIndexT  = tag $ Sub Dict
  entailEqT     = tag $ Sub Dict
  entailZTT     = tag $ Sub Dict
  entailNFDataT = tag $ Sub Dict
  entailRandomT = tag $ Sub Dict
  entailShowT   = tag $ Sub Dict
  entailModuleT = tag $ Sub Dict

  scalarPow = HT . scalarPow

  l       = onGPU OpL       l
  lInv    = onGPU OpLInv    lInv
  mulGPow = onGPU OpMulGPow mulGPow
  mulGDec = onGPU OpMulGDec mulGDec
  divGPow = onGPUMaybe OpDivGPow divGPow
  divGDec = onGPUMaybe OpDivGDec divGDec

  -- (scalarCRT, mulGCRT, divGCRT, crt, crtInv); Nothing when some modulus has no CRT basis —
  -- for Z_q types that is `lolhip_plan_has_crt`, which applies CT's own criterion (crtInfo)
  crtFuncs :: forall mon m r . (CRTrans mon r, Fact m, TElt HT r)
           => mon (r -> HT m r, HT m r -> HT m r, HT m r -> HT m r, HT m r -> HT m r, HT m r -> HT m r)
  crtFuncs = do
    (sc, mg, dg, c, ci) <- crtFuncs            -- CT's five, also the fall-through implementations
    return ( HT . sc
           , onGPU OpMulGCRT mg, onGPU OpDivGCRT dg
           , onGPU OpCRT c, onGPU OpCRTInv ci )

  tGaussianDec v = HT <$> tGaussianDec v       -- Double elements: CPU (GPU form: lolhip_gaussian_dec_batch, HIP.Batch)
  gSqNormDec (HT x) = gSqNormDec x

  twacePowDec = viaExt ExtTwacePowDec twacePowDec False
  embedPow    = viaExt ExtEmbedPow    embedPow    True
  embedDec    = viaExt ExtEmbedDec    embedDec    True

  crtExtFuncs = do
    (tw, em) <- crtExtFuncs
    return (viaExt ExtTwaceCRT tw False, viaExt ExtEmbedCRT em True)

  coeffs (HT x) = HT <$> coeffs x              -- list-valued; the slab form is HIP.Batch.coeffsBatch
  powBasisPow   = (fmap HT) <$> powBasisPow
  crtSetDec     = (fmap HT) <$> crtSetDec

  fmapT f (HT x) = HT (fmapT f x)
  zipWithT f (HT a) (HT b) = HT (zipWithT f a b)
  unzipT (HT x) = let (a, b) = unzipT x in (HT a, HT b)

-- twace (to the subring, toHi = False) / embed (toHi = True) through the extension's index tables
viaExt :: forall m m' r a b . (m `Divides` m', HipElt r, TElt CT r)
       => ExtOp -> (CT a r -> CT b r) -> Bool -> HT a r -> HT b r
viaExt op cpu toHi (HT x) =
  case (,) <$> planOf (Proxy :: Proxy '(m, r)) <*> planOf (Proxy :: Proxy '(m', r)) of
    Nothing       -> HT (cpu x)
    Just (lo, hi) ->
      let t    = maybe 1 length (hipModuli (Proxy :: Proxy r))
          nOut = t * (if toHi then proxy totientFact (Proxy :: Proxy m') else proxy totientFact (Proxy :: Proxy m))
      in HT $ fromVec $ extHost (extFor lo hi) op 1 nOut (toVec x)
