"""XML round trip of the force definition -- the reference's own serialization test restated field by field
(serialization/tests/TestSerializeSlicedNonbondedForce.cpp:22-175), plus the reading rules of the proxy
(serialization/src/SlicedNonbondedForceProxy.cpp:103-162: version check, optional properties and their defaults, subset 0 not written)."""
import xml.etree.ElementTree as ET

import pytest


def _make(snb):
    # TestSerializeSlicedNonbondedForce.cpp:25-57
    force = snb.SlicedNonbondedForce(3)
    force.setForceGroup(3)
    force.setName("custom name")
    force.setNonbondedMethod(snb.SlicedNonbondedForce.CutoffPeriodic)
    force.setSwitchingDistance(1.5)
    force.setUseSwitchingFunction(True)
    force.setCutoffDistance(2.0)
    force.setEwaldErrorTolerance(1e-3)
    force.setReactionFieldDielectric(50.0)
    force.setUseDispersionCorrection(False)
    force.setExceptionsUsePeriodicBoundaryConditions(True)
    force.setIncludeDirectSpace(False)
    force.setPMEParameters(0.5, 3, 5, 7)
    force.setLJPMEParameters(0.8, 4, 6, 7)
    force.addParticle(1, 0.1, 0.01)
    force.addParticle(0.5, 0.2, 0.02)
    force.addParticle(-0.5, 0.3, 0.03)
    force.setParticleSubset(0, 1)
    force.setParticleSubset(1, 2)
    force.addException(0, 1, 2, 0.5, 0.1)
    force.addException(1, 2, 0.2, 0.4, 0.2)
    force.addGlobalParameter("scale1", 1.0)
    force.addGlobalParameter("scale2", 2.0)
    force.addParticleParameterOffset("scale1", 2, 1.5, 2.0, 2.5)
    force.addExceptionParameterOffset("scale2", 1, -0.1, -0.2, -0.3)
    force.addGlobalParameter("lambda", 0.5)
    force.addScalingParameter("lambda", 0, 1, True, True)
    force.addScalingParameter("lambda", 1, 1, False, True)
    force.addEnergyParameterDerivative("lambda")
    return force


def test_round_trip_field_by_field(snb):
    force = _make(snb)
    xml = snb.XmlSerializer.serialize(force, "Force")
    force2 = snb.XmlSerializer.deserialize(xml)
    # TestSerializeSlicedNonbondedForce.cpp:67-175
    for getter in ("getForceGroup", "getName", "getNonbondedMethod", "getSwitchingDistance", "getUseSwitchingFunction", "getCutoffDistance",
                   "getEwaldErrorTolerance", "getReactionFieldDielectric", "getUseDispersionCorrection", "getExceptionsUsePeriodicBoundaryConditions",
                   "getNumParticles", "getNumExceptions", "getNumGlobalParameters", "getNumParticleParameterOffsets", "getNumExceptionParameterOffsets",
                   "getIncludeDirectSpace", "getPMEParameters", "getLJPMEParameters", "getNumSubsets", "getNumScalingParameters",
                   "getNumEnergyParameterDerivatives", "getReciprocalSpaceForceGroup"):
        assert getattr(force, getter)() == getattr(force2, getter)(), getter
    for i in range(force.getNumGlobalParameters()):
        assert force.getGlobalParameterName(i) == force2.getGlobalParameterName(i)
        assert force.getGlobalParameterDefaultValue(i) == force2.getGlobalParameterDefaultValue(i)
    for i in range(force.getNumParticleParameterOffsets()):
        assert force.getParticleParameterOffset(i) == force2.getParticleParameterOffset(i)
    for i in range(force.getNumExceptionParameterOffsets()):
        assert force.getExceptionParameterOffset(i) == force2.getExceptionParameterOffset(i)
    for i in range(force.getNumParticles()):
        assert tuple(force.getParticleParameters(i)) == tuple(force2.getParticleParameters(i))
        assert force.getParticleSubset(i) == force2.getParticleSubset(i)
    for i in range(force.getNumExceptions()):
        assert tuple(force.getExceptionParameters(i)) == tuple(force2.getExceptionParameters(i))
    for i in range(force.getNumScalingParameters()):
        assert tuple(force.getScalingParameter(i)) == tuple(force2.getScalingParameter(i))
    for i in range(force.getNumEnergyParameterDerivatives()):
        assert force.getEnergyParameterDerivativeName(i) == force2.getEnergyParameterDerivativeName(i)
    # a second trip reproduces the document exactly
    assert snb.XmlSerializer.serialize(force2, "Force") == xml


def test_document_shape_follows_the_proxy(snb):
    root = ET.fromstring(snb.XmlSerializer.serialize(_make(snb), "Force"))
    assert root.tag == "Force" and root.get("type") == "SlicedNonbondedForce" and root.get("version") == "1"
    # SlicedNonbondedForceProxy.cpp:24-50: the property names of the node
    for key in ("numSubsets", "forceGroup", "name", "method", "cutoff", "useSwitchingFunction", "switchingDistance", "ewaldTolerance", "rfDielectric",
                "dispersionCorrection", "exceptionsUsePeriodic", "includeDirectSpace", "alpha", "nx", "ny", "nz", "ljAlpha", "ljnx", "ljny", "ljnz", "recipForceGroup"):
        assert root.get(key) is not None, key
    # :51-100: child nodes in the proxy's order, element names as it creates them
    assert [c.tag for c in root] == ["GlobalParameters", "ParticleOffsets", "ExceptionOffsets", "Particles", "Exceptions", "Subsets", "scalingParameters", "energyParameterDerivatives"]
    assert [c.tag for c in root.find("GlobalParameters")] == ["Parameter"] * 3
    assert root.find("ParticleOffsets")[0].attrib == {"parameter": "scale1", "particle": "2", "q": "1.5", "sig": "2.0", "eps": "2.5"}
    assert root.find("ExceptionOffsets")[0].get("exception") == "1"
    # only particles outside subset 0 are listed (:87-91)
    assert [(c.get("index"), c.get("subset")) for c in root.find("Subsets")] == [("0", "1"), ("1", "2")]
    assert root.find("scalingParameters")[1].attrib == {"parameter": "lambda", "subset1": "1", "subset2": "1", "includeCoulomb": "0", "includeLJ": "1"}
    assert root.find("energyParameterDerivatives")[0].tag == "energyParameterDerivative"


def test_reading_rules(snb):
    xml = snb.XmlSerializer.serialize(_make(snb))
    with pytest.raises(snb.OpenMMException):      # :104-106
        snb.XmlSerializer.deserialize(xml.replace('version="1"', 'version="2"'))
    # optional properties fall back to the proxy's defaults (:109-133)
    root = ET.fromstring(xml)
    for key in ("forceGroup", "name", "useSwitchingFunction", "switchingDistance", "includeDirectSpace", "alpha", "nx", "ny", "nz", "ljAlpha", "ljnx", "ljny", "ljnz", "recipForceGroup"):
        del root.attrib[key]
    f = snb.XmlSerializer.deserialize(ET.tostring(root, encoding="unicode"))
    assert f.getForceGroup() == 0 and f.getName() == "SlicedNonbondedForce" and f.getUseSwitchingFunction() is False and f.getSwitchingDistance() == -1.0
    assert f.getIncludeDirectSpace() is True and f.getPMEParameters() == (0.0, 0, 0, 0) and f.getLJPMEParameters() == (0.0, 0, 0, 0) and f.getReciprocalSpaceForceGroup() == -1
    # a mandatory property or child node that is missing is an error
    root = ET.fromstring(xml); del root.attrib["cutoff"]
    with pytest.raises(snb.OpenMMException):
        snb.XmlSerializer.deserialize(ET.tostring(root, encoding="unicode"))
    root = ET.fromstring(xml); root.remove(root.find("Subsets"))
    with pytest.raises(snb.OpenMMException):
        snb.XmlSerializer.deserialize(ET.tostring(root, encoding="unicode"))
