"""XML form of a SlicedNonbondedForce definition (saved systems keep working).

Mirrors the reference's serialization proxy -- property and child-node names, the version number, the defaults applied on
reading and the "only non-zero subsets are written" rule follow serialization/src/SlicedNonbondedForceProxy.cpp:23-101
(write) and :103-162 (read); the type name "SlicedNonbondedForce" is the one the proxy registers (:20).  The surrounding
document syntax (root element named by the caller, `type` / `version` / `openmmVersion` attributes, booleans as 0/1) is that
of OpenMM's XmlSerializer, a third-party library absent from /root/reference: restated from its published format, so byte
equality with OpenMM's writer is NOT pinned -- what is pinned (tests/test_serialization.py) is the field-by-field round trip
of the reference's own serialization test (serialization/tests/TestSerializeSlicedNonbondedForce.cpp:22-175).
No arithmetic lives here.
"""
from __future__ import annotations

import xml.etree.ElementTree as ET

from .force import OpenMMException, SlicedNonbondedForce

TYPE_NAME = "SlicedNonbondedForce"
VERSION = 1


def _d(x) -> str:
    return repr(float(x))


def _b(x) -> str:
    return "1" if x else "0"


def serialize(force: SlicedNonbondedForce, root_name: str = "Force") -> str:
    """The force as an XML document (SlicedNonbondedForceProxy.cpp:23-101)."""
    root = ET.Element(root_name)
    root.set("type", TYPE_NAME)
    root.set("version", str(VERSION))
    root.set("numSubsets", str(force.getNumSubsets()))
    root.set("forceGroup", str(force.getForceGroup()))
    root.set("name", force.getName())
    root.set("method", str(int(force.getNonbondedMethod())))
    root.set("cutoff", _d(force.getCutoffDistance()))
    root.set("useSwitchingFunction", _b(force.getUseSwitchingFunction()))
    root.set("switchingDistance", _d(force.getSwitchingDistance()))
    root.set("ewaldTolerance", _d(force.getEwaldErrorTolerance()))
    root.set("rfDielectric", _d(force.getReactionFieldDielectric()))
    root.set("dispersionCorrection", _b(force.getUseDispersionCorrection()))
    root.set("exceptionsUsePeriodic", _b(force.getExceptionsUsePeriodicBoundaryConditions()))
    root.set("includeDirectSpace", _b(force.getIncludeDirectSpace()))
    alpha, nx, ny, nz = force.getPMEParameters()
    root.set("alpha", _d(alpha)); root.set("nx", str(nx)); root.set("ny", str(ny)); root.set("nz", str(nz))
    alpha, nx, ny, nz = force.getLJPMEParameters()
    root.set("ljAlpha", _d(alpha)); root.set("ljnx", str(nx)); root.set("ljny", str(ny)); root.set("ljnz", str(nz))
    root.set("recipForceGroup", str(force.getReciprocalSpaceForceGroup()))
    node = ET.SubElement(root, "GlobalParameters")
    for i in range(force.getNumGlobalParameters()):
        ET.SubElement(node, "Parameter", name=force.getGlobalParameterName(i), default=_d(force.getGlobalParameterDefaultValue(i)))
    node = ET.SubElement(root, "ParticleOffsets")
    for i in range(force.getNumParticleParameterOffsets()):
        parameter, particle, q, sig, eps = force.getParticleParameterOffset(i)
        ET.SubElement(node, "Offset", parameter=parameter, particle=str(particle), q=_d(q), sig=_d(sig), eps=_d(eps))
    node = ET.SubElement(root, "ExceptionOffsets")
    for i in range(force.getNumExceptionParameterOffsets()):
        parameter, exception, q, sig, eps = force.getExceptionParameterOffset(i)
        ET.SubElement(node, "Offset", parameter=parameter, exception=str(exception), q=_d(q), sig=_d(sig), eps=_d(eps))
    node = ET.SubElement(root, "Particles")
    for i in range(force.getNumParticles()):
        q, sig, eps = force.getParticleParameters(i)
        ET.SubElement(node, "Particle", q=_d(q), sig=_d(sig), eps=_d(eps))
    node = ET.SubElement(root, "Exceptions")
    for i in range(force.getNumExceptions()):
        p1, p2, q, sig, eps = force.getExceptionParameters(i)
        ET.SubElement(node, "Exception", p1=str(p1), p2=str(p2), q=_d(q), sig=_d(sig), eps=_d(eps))
    node = ET.SubElement(root, "Subsets")
    for i in range(force.getNumParticles()):
        subset = force.getParticleSubset(i)
        if subset != 0:      # subset 0 is the default and is not written (:87-91)
            ET.SubElement(node, "Subset", index=str(i), subset=str(subset))
    node = ET.SubElement(root, "scalingParameters")
    for i in range(force.getNumScalingParameters()):
        parameter, s1, s2, coulomb, lj = force.getScalingParameter(i)
        ET.SubElement(node, "scalingParameter", parameter=parameter, subset1=str(s1), subset2=str(s2), includeCoulomb=_b(coulomb), includeLJ=_b(lj))
    node = ET.SubElement(root, "energyParameterDerivatives")
    for i in range(force.getNumEnergyParameterDerivatives()):
        ET.SubElement(node, "energyParameterDerivative", parameter=force.getEnergyParameterDerivativeName(i))
    ET.indent(root, space="\t")
    return '<?xml version="1.0" ?>\n' + ET.tostring(root, encoding="unicode") + "\n"


def _children(root, name):
    node = root.find(name)
    if node is None:
        raise OpenMMException("Unknown child node '%s'" % name)      # SerializationNode::getChildNode throws for a missing child
    return list(node)


def _bool(text: str) -> bool:
    return text.strip().lower() not in ("0", "false", "")


def deserialize(xml: str) -> SlicedNonbondedForce:
    """Rebuild the force (SlicedNonbondedForceProxy.cpp:103-162): version 1 only; optional properties take the proxy's defaults."""
    root = ET.fromstring(xml)
    if root.get("type", TYPE_NAME) != TYPE_NAME:
        raise OpenMMException("Unknown object type '%s'" % root.get("type"))

    def need(key):
        if root.get(key) is None:
            raise OpenMMException("Unknown property '%s'" % key)
        return root.get(key)

    if int(need("version")) != VERSION:
        raise OpenMMException("Unsupported version number")
    force = SlicedNonbondedForce(int(need("numSubsets")))
    force.setForceGroup(int(root.get("forceGroup", "0")))
    force.setName(root.get("name", force.getName()))
    force.setNonbondedMethod(int(need("method")))
    force.setCutoffDistance(float(need("cutoff")))
    force.setUseSwitchingFunction(_bool(root.get("useSwitchingFunction", "0")))
    force.setSwitchingDistance(float(root.get("switchingDistance", "-1.0")))
    force.setEwaldErrorTolerance(float(need("ewaldTolerance")))
    force.setReactionFieldDielectric(float(need("rfDielectric")))
    force.setUseDispersionCorrection(_bool(need("dispersionCorrection")))
    if root.get("includeDirectSpace") is not None:
        force.setIncludeDirectSpace(_bool(root.get("includeDirectSpace")))
    force.setPMEParameters(float(root.get("alpha", "0.0")), int(root.get("nx", "0")), int(root.get("ny", "0")), int(root.get("nz", "0")))
    force.setLJPMEParameters(float(root.get("ljAlpha", "0.0")), int(root.get("ljnx", "0")), int(root.get("ljny", "0")), int(root.get("ljnz", "0")))
    force.setReciprocalSpaceForceGroup(int(root.get("recipForceGroup", "-1")))
    for n in _children(root, "GlobalParameters"):
        force.addGlobalParameter(n.get("name"), float(n.get("default")))
    for n in _children(root, "ParticleOffsets"):
        force.addParticleParameterOffset(n.get("parameter"), int(n.get("particle")), float(n.get("q")), float(n.get("sig")), float(n.get("eps")))
    for n in _children(root, "ExceptionOffsets"):
        force.addExceptionParameterOffset(n.get("parameter"), int(n.get("exception")), float(n.get("q")), float(n.get("sig")), float(n.get("eps")))
    force.setExceptionsUsePeriodicBoundaryConditions(_bool(need("exceptionsUsePeriodic")))
    for n in _children(root, "Particles"):
        force.addParticle(float(n.get("q")), float(n.get("sig")), float(n.get("eps")))
    for n in _children(root, "Exceptions"):
        force.addException(int(n.get("p1")), int(n.get("p2")), float(n.get("q")), float(n.get("sig")), float(n.get("eps")))
    for n in _children(root, "Subsets"):
        force.setParticleSubset(int(n.get("index")), int(n.get("subset")))
    for n in _children(root, "scalingParameters"):
        force.addScalingParameter(n.get("parameter"), int(n.get("subset1")), int(n.get("subset2")), _bool(n.get("includeCoulomb")), _bool(n.get("includeLJ")))
    for n in _children(root, "energyParameterDerivatives"):
        force.addEnergyParameterDerivative(n.get("parameter"))
    return force


class XmlSerializer:
    """`XmlSerializer.serialize(force)` / `XmlSerializer.deserialize(xml)`: the calls user scripts make on OpenMM's class of that name."""
    serialize = staticmethod(serialize)
    deserialize = staticmethod(deserialize)
